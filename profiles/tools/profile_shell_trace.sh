cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-v4_shell}; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --workload shell --steps 5 --warmup 2 --no-cpu-baseline > $O/kt.log 2>&1
f=$(find $O/kt -name "*.db" | head -1); python profiles/summarize_rocpd.py "$f" > $O/kt.txt 2>&1
find $O -name "*.db" -delete; rm -rf $O/kt
tail -1 $O/kt.log | cut -c1-200; head -14 $O/kt.txt | cut -c1-70,105-170
