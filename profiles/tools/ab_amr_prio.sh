#!/bin/bash
# same-box A/B of stream priorities for the children-beside-far-boxes schedule (bench.py --workload amr)
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/ab_amr_prio.txt
python -c "import torch; print('priority_range', torch.cuda.Stream.priority_range())" > $out
for rep in 1 2; do
  for cfg in "0 low -" "1 normal -" "1 low -" "1 low high" "1 normal high"; do
    set -- $cfg
    mp=""; [ "$3" != "-" ] && mp=$3
    QK_AMR_OVERLAP=$1 QK_AMR_FAR_PRIORITY=$2 QK_AMR_MAIN_PRIORITY=$mp python bench.py --workload amr --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('overlap=$1 far=$2 main=$3 rep=$rep value=%.1f ms_per_step=%.3f dE=%.2e %s' % (d['value'], d['ms_per_step'], d['config']['composite_energy_relative_change'], d['config']['children_beside_far_boxes']))" >> $out
  done
done
cat $out
