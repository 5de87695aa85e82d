#!/bin/bash
# same-box A/B of AmrSimulation.overlap_children on bench.py --workload amr (config 5 geometry, one GPU): three runs each, interleaved
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/ab_amr_overlap.txt
: > $out
for rep in 1 2 3; do
  for ov in 0 1; do
    QK_AMR_OVERLAP=$ov python bench.py --workload amr --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('overlap=$ov rep=$rep value=%.1f ms_per_step=%.3f dE=%.2e %s' % (d['value'], d['ms_per_step'], d['config']['composite_energy_relative_change'], d['config']['children_beside_far_boxes']))" >> $out
  done
done
cat $out
