# kernel trace of the Sedov AMR workload: bash profiles/tools/profile_amr_trace.sh  -> gpurun_out/v4_amr/kt.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-v4_amr}; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --workload amr --steps 20 --warmup 5 --no-cpu-baseline > $O/kt.log 2>&1
f=$(find $O/kt -name "*.db" | head -1); python profiles/summarize_rocpd.py "$f" > $O/kt.txt 2>&1
find $O -name "*.db" -delete; rm -rf $O/kt
tail -1 $O/kt.log | cut -c1-200; head -26 $O/kt.txt | cut -c1-70,105-170
