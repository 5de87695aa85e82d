cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof2
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/write.log 2>&1
for d in kt fetch write; do f=$(find $O/$d -name "*.db" | head -1); python profiles/summarize_rocpd.py $f > $O/$d.txt 2>&1; done
find $O -name "*.db" -delete
tail -2 $O/kt.log | cut -c1-300
head -20 $O/kt.txt
