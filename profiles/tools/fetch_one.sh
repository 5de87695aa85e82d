# FETCH_SIZE of the kernels of one short bench run for a given library: fetch_one.sh <libname>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/fetch_$1; mkdir -p $O
QK_LIB_PATH=$R/quokka_amd/lib/libqk_$1.so rocprofv3 --pmc FETCH_SIZE -d $O/f -- python bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 > $O/log 2>&1
f=$(find $O/f -name "*.db" | head -1); python profiles/summarize_rocpd.py "$f" | grep -A8 "^# counters" | grep "pre3\|sweep_x" | awk '{print $1, $2, $(NF)}'
rm -rf $O/f
