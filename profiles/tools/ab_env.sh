# usage: ab_env.sh VAR v1 v2 ... : the default bench with environment variable VAR set to each value in turn, twice
VAR=$1; shift
for r in 1 2; do
for v in "$@"; do
  env $VAR=$v python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>&1 | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_ms_per_launch']; print('$VAR=$v', round(d['value'],1), {n: round(t,3) for n,t in k.items()})"
done
done
