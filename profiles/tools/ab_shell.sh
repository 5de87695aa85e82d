# A/B builds of libquokka_amd.so on the RadhydroShell workload, same box (see ab_bench.sh):
#   gpurun -- 'bash profiles/tools/ab_shell.sh A B'   with quokka_amd/lib/libqk_<name>.so
for r in 1 2; do
  for v in "$@"; do
    QK_LIB_PATH=$PWD/quokka_amd/lib/libqk_$v.so python bench.py --workload shell --steps 4 --warmup 1 2>&1 | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_launch']; print('$v', round(d['value'],1), {n: round(t,3) for n,t in k.items() if n.startswith('rad')})"
  done
done
