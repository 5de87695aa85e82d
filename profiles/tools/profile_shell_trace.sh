cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/shell_kt; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --workload shell --steps 4 --warmup 2 --no-cpu-baseline > $O/kt.log 2>&1
f=$(find $O/kt -name "*.db" | head -1); python profiles/summarize_rocpd.py "$f" > $O/kt.txt 2>&1
rm -rf $O/kt
tail -1 $O/kt.log | cut -c1-200; head -24 $O/kt.txt | cut -c1-70,105-175
