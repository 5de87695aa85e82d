#!/bin/bash
# same-box A/B of the pre-pass with the next plane's values requested one plane ahead (QK_PRE_PREFETCH: 0 none, 1 both stages, 2 only the stage that
# reads primitives), three library builds (quokka_amd/lib/libquokka_amd{,_pf1,_pf2}.so via QK_LIB_PATH), headline configuration, interleaved
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/ab_pre_prefetch.txt
: > $out
for n in 256 512; do
  for rep in 1 2 3; do
    for v in "" _pf1 _pf2; do
      QK_LIB_PATH=$PWD/quokka_amd/lib/libquokka_amd$v.so python bench.py --ncell $n --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k={a:round(b,4) for a,b in d['roofline'].get('all_kernels_ms_per_launch',{}).items()}
print('n=$n lib=base$v rep=$rep value=%.1f ms_per_step=%.3f k_pre=%.4f kernels=%s' % (d['value'], d['ms_per_step'], k.get('k_pre',0), json.dumps(k)))" >> $out
    done
  done
done
cat $out
