#!/bin/bash
# same-box A/B of several library builds on the headline: bash profiles/tools/ab_libs_many.sh <ncell> <lib.so> ...   (QK_LIB_PATH; three interleaved rounds)
cd "$(dirname "$0")/../.." || exit 1
n=$1; shift
mkdir -p gpurun_out
out=gpurun_out/ab_libs_many.txt
: > $out
for rep in 1 2 3; do
  for lib in "$@"; do
    QK_LIB_PATH=$PWD/$lib python bench.py --ncell $n --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k={a:round(b,4) for a,b in d['roofline'].get('all_kernels_ms_per_launch',{}).items() if a.startswith('k_')}
print('n=$n lib=$(basename $lib) rep=$rep value=%.1f ms_per_step=%.3f kernels=%s' % (d['value'], d['ms_per_step'], json.dumps(k)))" >> $out
  done
done
cat $out
