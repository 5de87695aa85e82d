#!/bin/bash
# ab_shell_variants.sh name1 name2 ...: RadhydroShell 256^3 (bench.py --workload shell) under library variants ("base" = the tree's library)
for rep in 1 2; do
  for v in "$@"; do
    lib=quokka_amd/lib/variants/libqk_$v.so; [ "$v" = base ] && lib=quokka_amd/lib/libquokka_amd.so
    QK_LIB_PATH=$PWD/$lib python bench.py --workload shell --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
k = d.get('kernels_ms_per_launch') or {}
print('shell256 %-10s rep $rep value %.1f M  ms/step %.2f ' % ('$v', d['value'], d['ms_per_step']), {a: round(b, 4) for a, b in sorted(k.items()) if 'rad' in a})
"
  done
done
