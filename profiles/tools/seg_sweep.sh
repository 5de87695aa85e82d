for seg in 1 2 4 8; do
QK_MARCH_SEGMENTS=$seg python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_ms_per_launch']; print('seg $seg', round(d['value'],1), {n: round(t,3) for n,t in k.items()})"
done
