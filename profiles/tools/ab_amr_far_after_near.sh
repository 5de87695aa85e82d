#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
out=gpurun_out/ab_amr_far_after_near.txt; : > $out
for rep in 1 2 3; do for v in 0 1; do
QK_AMR_FAR_AFTER_NEAR=$v python bench.py --workload amr --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('far_after_near=$v rep=$rep value=%.1f ms_per_step=%.3f %s' % (d['value'], d['ms_per_step'], d['config']['children_beside_far_boxes']))" >> $out
done; done; cat $out
