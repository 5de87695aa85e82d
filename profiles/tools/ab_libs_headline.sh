#!/bin/bash
# same-box A/B of two library builds on the headline: bash profiles/tools/ab_libs_headline.sh <libA.so> <libB.so> [ncell ...]   (QK_LIB_PATH)
cd "$(dirname "$0")/../.." || exit 1
A=$1; B=$2; shift 2
mkdir -p gpurun_out
out=gpurun_out/ab_libs_headline.txt
: > $out
for n in ${@:-256}; do
  for rep in 1 2 3; do
    for lib in $A $B; do
      QK_LIB_PATH=$PWD/$lib python bench.py --ncell $n --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k={a:round(b,4) for a,b in d['roofline'].get('all_kernels_ms_per_launch',{}).items()}
print('n=$n lib=$lib rep=$rep value=%.1f ms_per_step=%.3f kernels=%s' % (d['value'], d['ms_per_step'], json.dumps(k)))" >> $out
    done
  done
done
cat $out
