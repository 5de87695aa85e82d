# usage: ab_env_shell.sh VAR v1 v2 ... : the RadhydroShell 256^3 workload with environment variable VAR set to each value in turn, twice (same box)
VAR=$1; shift
for r in 1 2; do
for v in "$@"; do
  env $VAR=$v python bench.py --workload shell --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernels_ms_per_launch',{}); print('$VAR=$v', round(d['value'],1), round(d['ms_per_step'],2), {n: round(t,3) for n,t in k.items() if n.startswith('rad')})"
done
done
