# timeline of one coarse step of bench.py --workload amr: bash profiles/tools/profile_amr_timeline.sh TAG [QK_AMR_OVERLAP]  -> gpurun_out/TAG/timeline.txt
tag=${1:-amr_tl}; ov=${2:-1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
(cd $R && QK_AMR_OVERLAP=$ov rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python bench.py --workload amr --steps 16 --warmup 4 --no-cpu-baseline > $O/run.log 2>&1)
python $R/profiles/tools/timeline.py $O/kt 12 600 > $O/timeline.txt 2>&1
rm -rf $O/kt
tail -1 $O/run.log | cut -c1-120; head -5 $O/timeline.txt
