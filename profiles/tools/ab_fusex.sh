#!/bin/bash
# same-box A/B of the X sweep folded into the Y march (QK_FUSEX=1) against the four-kernel stage: headline + per-kernel times at 256^3 and 512^3
out=gpurun_out/ab_fusex.txt
: > $out
for n in 256 512; do
  for rep in 1 2; do
    for f in 0 1; do
      QK_FUSEX=$f python bench.py --ncell $n --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
k = (d.get('roofline') or {}).get('all_kernels_ms_per_launch') or {}
print('ncell $n fusex $f rep $rep value %.1f M  ms/step %.3f  fofc %s retries %s' % (d['value'], d['ms_per_step'], d['config'].get('fofc_stages'), d['config'].get('retries')), {a: round(b, 4) for a, b in sorted(k.items())} if k else [x for x in d.keys()])
" >> $out
    done
  done
done
cat $out
