#!/bin/bash
# same-box A/B of an environment knob on bench.py's headline: r4_ab_env.sh OUTDIR VAR "v1 v2 ..." [extra bench args]
# (each value runs twice, interleaved, at 256^3 and once at 512^3; prints value / ms per step / per-kernel ms)
out=$1; var=$2; vals=$3; shift 3
mkdir -p "$out"
for rep in 1 2; do
  for v in $vals; do
    env $var=$v python bench.py --no-secondary --no-cpu-baseline "$@" > "$out/ab_${var}_${v}_256_$rep.json" 2> "$out/ab_${var}_${v}_256_$rep.err"
  done
done
for v in $vals; do
  env $var=$v python bench.py --no-secondary --no-cpu-baseline --ncell 512 --steps 8 --warmup 2 "$@" > "$out/ab_${var}_${v}_512.json" 2> "$out/ab_${var}_${v}_512.err"
done
python - "$out" "$var" <<'PY'
import glob, json, os, sys
out, var = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(os.path.join(out, f"ab_{var}_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "FAILED", e, open(f.replace(".json", ".err")).read()[-400:]); continue
    k = d["roofline"]["all_kernels_ms_per_launch"]
    print(f"{os.path.basename(f):40s} {d['value']:8.1f} M  {d['ms_per_step']:7.3f} ms  " + "  ".join(f"{n}={v:.3f}" for n, v in k.items()))
PY
