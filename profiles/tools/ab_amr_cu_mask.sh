#!/bin/bash
# same-box A/B of the CU mask of the far boxes' stream (QK_AMR_FAR_CU_MASK: 0 no mask, N every N-th CU kept free) on bench.py --workload amr
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/ab_amr_cu_mask.txt
: > $out
for rep in 1 2 3; do
  for m in 0 8 4 16; do
    QK_AMR_FAR_CU_MASK=$m python bench.py --workload amr --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('cu_mask=$m rep=$rep value=%.1f ms_per_step=%.3f dE=%.2e %s' % (d['value'], d['ms_per_step'], d['config']['composite_energy_relative_change'], d['config']['children_beside_far_boxes']))" >> $out
  done
done
cat $out
