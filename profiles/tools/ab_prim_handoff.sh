#!/bin/bash
# same-box A/B of the primitive hand-off between the stages of a step (QK_PRIM_HANDOFF; qk_hydro_stage_args::prim_out / prim_in) on the headline
# (Sedov 256^3, carried form) and at 512^3: interleaved runs, each line with the per-kernel HIP-event times of the line
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/ab_prim_handoff.txt
: > $out
for n in 256 512; do
  for rep in 1 2 3; do
    for ph in 0 1; do
      QK_PRIM_HANDOFF=$ph python bench.py --ncell $n --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k={a:round(b,4) for a,b in d['roofline'].get('all_kernels_ms_per_launch',{}).items()}
print('n=$n handoff=$ph rep=$rep value=%.1f ms_per_step=%.3f frac=%.4f kernels=%s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], json.dumps(k)))" >> $out
    done
  done
done
cat $out
