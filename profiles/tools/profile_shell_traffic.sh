# HBM traffic of the RadhydroShell kernels: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --workload shell`
# usage: bash profiles/tools/profile_shell_traffic.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-shellpmc}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -- python bench.py --workload shell --steps 2 --warmup 1 > $O/fetch.log 2>&1 < /dev/null
rocprofv3 --pmc WRITE_SIZE -d $O/write -- python bench.py --workload shell --steps 2 --warmup 1 > $O/write.log 2>&1 < /dev/null
for d in fetch write; do
  f=$(find $O/$d -name "*.db" | head -1)
  if [ -n "$f" ]; then python profiles/summarize_rocpd.py "$f" > $O/$d.txt 2>&1; fi
done
find $O -name "*.db" -delete
grep -E "k_rad_cells" $O/fetch.txt | grep FETCH | cut -c1-70,110-190
grep -E "k_rad_cells" $O/write.txt | grep WRITE | cut -c1-70,110-190
