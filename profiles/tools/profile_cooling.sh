# rocprofv3 summaries of the cooling block (python bench.py --cooling-only): kernel trace and one SQ pass as SEPARATE runs.
# usage: bash profiles/tools/profile_cooling.sh <tag>     -> gpurun_out/<tag>/{kt,sq1}.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-cool}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
B="python bench.py --cooling-only"
rocprofv3 --kernel-trace --stats -d $O/kt -- $B > $O/kt.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/sq1 -- $B > $O/sq1.log 2>&1
for d in kt sq1; do
  f=$(find $O/$d -name "*.db" | head -1)
  if [ -n "$f" ]; then python profiles/summarize_rocpd.py "$f" > $O/$d.txt 2>&1; fi
done
find $O -name "*.db" -delete
rm -rf $O/kt $O/sq1
head -8 $O/kt.txt | cut -c1-70,105-175
grep -i cooling $O/sq1.txt | head -12
