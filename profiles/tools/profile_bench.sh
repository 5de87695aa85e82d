# rocprofv3 summaries of the default bench command: kernel trace, FETCH_SIZE, WRITE_SIZE and two SQ passes as SEPARATE runs
# (PMC passes never share a run with --kernel-trace; summaries via profiles/summarize_rocpd.py).
# usage: bash profiles/tools/profile_bench.sh <tag> [ncell]     -> gpurun_out/<tag>/{kt,fetch,write,sq1,sq2}.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-prof}
N=${2:-256}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
B="python bench.py --no-cpu-baseline --no-secondary --ncell $N"
rocprofv3 --kernel-trace --stats -d $O/kt -- $B --steps 10 --warmup 3 > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -- $B --steps 4 --warmup 1 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -- $B --steps 4 --warmup 1 > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/sq1 -- $B --steps 3 --warmup 1 > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $O/sq2 -- $B --steps 3 --warmup 1 > $O/sq2.log 2>&1
for d in kt fetch write sq1 sq2; do
  f=$(find $O/$d -name "*.db" | head -1)
  if [ -n "$f" ]; then python profiles/summarize_rocpd.py "$f" > $O/$d.txt 2>&1; fi
done
find $O -name "*.db" -delete
rm -rf $O/kt $O/fetch $O/write $O/sq1 $O/sq2
tail -1 $O/kt.log | cut -c1-300
head -12 $O/kt.txt | cut -c1-60,105-175
