# same-box A/B of library variants at 512^3: bash profiles/tools/ab_pre512.sh <variant> <variant> ...   (quokka_amd/lib/libqk_<variant>.so)
for v in "$@"; do
  QK_LIB_PATH=$PWD/quokka_amd/lib/libqk_$v.so python bench.py --ncell 512 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>&1 | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_ms_per_launch']; print('512 $v', round(d['value'],1), {n: round(t,3) for n,t in k.items()})"
done
