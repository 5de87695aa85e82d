# same-box A/B of library variants on the Sedov AMR workload: bash profiles/tools/ab_amr.sh <variant> ...   (quokka_amd/lib/libqk_<variant>.so)
for r in 1 2; do
for v in "$@"; do
  QK_LIB_PATH=$PWD/quokka_amd/lib/libqk_$v.so python bench.py --workload amr --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('amr $v', round(d['value'],1), round(d['ms_per_step'],3))"
done
done
