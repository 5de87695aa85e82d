#!/bin/bash
# ab_variants.sh NCELL name1 name2 ...: headline + kernel times of library variants (quokka_amd/lib/variants/libqk_NAME.so; "base" = the tree's library)
n=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    lib=quokka_amd/lib/variants/libqk_$v.so; [ "$v" = base ] && lib=quokka_amd/lib/libquokka_amd.so
    QK_LIB_PATH=$PWD/$lib python bench.py --ncell $n --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
k = (d.get('roofline') or {}).get('all_kernels_ms_per_launch') or {}
print('ncell $n %-12s rep $rep value %.1f M  ms/step %.3f ' % ('$v', d['value'], d['ms_per_step']), {a: round(b, 4) for a, b in sorted(k.items()) if a.startswith('k_')})
"
  done
done
