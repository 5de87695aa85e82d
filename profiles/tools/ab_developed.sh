#!/bin/bash
# developed-flow block under library variants / march segment counts: value + kernel times
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --developed-only --steps 50 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())['developed']
print('%-28s value %.1f M  ms/step %.3f ' % ('$label', d['value'], d['ms_per_step']), {a: round(b, 4) for a, b in sorted(d['kernels_ms_per_launch'].items()) if a.startswith('k_')})
"
}
for rep in 1 2; do
  run base X=1
  run mb2 QK_LIB_PATH=$PWD/quokka_amd/lib/variants/libqk_mb2.so
  run mb1 QK_LIB_PATH=$PWD/quokka_amd/lib/variants/libqk_mb1.so
  run seg2 QK_MARCH_SEGMENTS=2
  run seg2_mb2 QK_MARCH_SEGMENTS=2 QK_LIB_PATH=$PWD/quokka_amd/lib/variants/libqk_mb2.so
done
