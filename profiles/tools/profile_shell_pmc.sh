# usage: profile_shell_pmc.sh "<counters>"  — one rocprofv3 --pmc pass over one step of the radhydro shell bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_shell
mkdir -p $O
cd $R
rocprofv3 --pmc $1 -d $O/pmc -- python bench.py --workload shell --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc.log 2>&1
f=$(find $O/pmc -name "*.db" | head -1); python profiles/summarize_rocpd.py $f > $O/pmc_$2.txt 2>&1
find $O -name "*.db" -delete
grep -A40 "counters" $O/pmc_$2.txt | grep -i "rad_\|Source\|counters" | cut -c1-60,100-200 | head -40
