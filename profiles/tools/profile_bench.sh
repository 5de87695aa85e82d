# rocprofv3 summaries of the default bench command (python bench.py): kernel trace + FETCH_SIZE + WRITE_SIZE as three separate runs
# (PMC passes never share a run with --kernel-trace; summaries via profiles/summarize_rocpd.py).  usage: bash profiles/tools/profile_bench.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-prof}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --no-cpu-baseline > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/write.log 2>&1
for d in kt fetch write; do
  f=$(find $O/$d -name "*.db" | head -1)
  if [ -n "$f" ]; then python profiles/summarize_rocpd.py "$f" > $O/$d.txt 2>&1; fi
done
find $O -name "*.db" -delete
tail -1 $O/kt.log | cut -c1-400
head -14 $O/kt.txt | cut -c1-60,105-175
