#!/bin/bash
# same-box A/B of QK_AMR_FAR_SPLIT (the far boxes of level 0 as n launches in a row on the side stream) on bench.py --workload amr
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/ab_amr_far_split.txt
: > $out
for rep in 1 2 3; do
  for m in 1 2 3 7; do
    QK_AMR_FAR_SPLIT=$m python bench.py --workload amr --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('far_split=$m rep=$rep value=%.1f ms_per_step=%.3f dE=%.2e' % (d['value'], d['ms_per_step'], d['config']['composite_energy_relative_change']))" >> $out
  done
done
cat $out
