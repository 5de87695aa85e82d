#!/bin/bash
# same-box A/B of library builds quokka_amd/lib/libqk_<name>.so on bench.py --workload amr and on the headline: bash profiles/tools/ab_amr_libs.sh A B ...
cd "$(dirname "$0")/../.." || exit 1
out=gpurun_out/ab_amr_libs.txt; : > $out
for rep in 1 2 3; do for v in "$@"; do
QK_LIB_PATH=$PWD/quokka_amd/lib/libqk_$v.so python bench.py --workload amr --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('amr  lib=$v rep=$rep value=%.1f ms_per_step=%.3f' % (d['value'], d['ms_per_step']))" >> $out
done; done
for rep in 1 2; do for v in "$@"; do
QK_LIB_PATH=$PWD/quokka_amd/lib/libqk_$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['all_kernels_ms_per_launch']
print('sedov256 lib=$v rep=$rep value=%.1f' % d['value'], {n: round(t,3) for n,t in k.items()})" >> $out
done; done
cat $out
