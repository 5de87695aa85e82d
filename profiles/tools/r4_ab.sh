#!/bin/bash
# same-box A/B of builds / environment knobs on bench.py's headline: r4_ab.sh OUTDIR [bench args --] "label|VAR=a VAR2=b" ...
# every variant runs twice at 256^3 (interleaved) and once at 512^3; prints value / ms per step / per-kernel ms
out=$1; shift
extra=""
if [ "$1" = "--args" ]; then extra=$2; shift 2; fi
mkdir -p "$out"
for rep in 1 2; do
  for spec in "$@"; do
    label=${spec%%|*}; envs=${spec#*|}
    env $envs python bench.py --no-secondary --no-cpu-baseline $extra > "$out/ab_${label}_256_$rep.json" 2> "$out/ab_${label}_256_$rep.err"
  done
done
for spec in "$@"; do
  label=${spec%%|*}; envs=${spec#*|}
  env $envs python bench.py --no-secondary --no-cpu-baseline --ncell 512 --steps 8 --warmup 2 $extra > "$out/ab_${label}_512.json" 2> "$out/ab_${label}_512.err"
done
python - "$out" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "ab_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "FAILED", e, open(f.replace(".json", ".err")).read()[-400:]); continue
    k = d["roofline"]["all_kernels_ms_per_launch"]
    print(f"{os.path.basename(f):34s} {d['value']:8.1f} M  {d['ms_per_step']:7.3f} ms  " + "  ".join(f"{n.replace('k_sweep_','').replace('ghost_','g_')}={v:.3f}" for n, v in k.items()))
PY
