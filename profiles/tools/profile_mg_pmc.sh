# usage: profile_mg_pmc.sh "<counters>" <tag> — one rocprofv3 --pmc pass over the multigroup timing problem (mg_kernel_time.py, smaller: 128 x 64 x 64, 1 step)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_mg
mkdir -p $O
cd $R
timeout 600 rocprofv3 --pmc $1 -d $O/pmc -- python profiles/tools/mg_kernel_time.py 128 64 1 > $O/pmc.log 2>&1
f=$(find $O/pmc -name "*.db" | head -1); python profiles/summarize_rocpd.py $f > $O/pmc_$2.txt 2>&1
find $O -name "*.db" -delete
grep -A60 "counters" $O/pmc_$2.txt | grep -i "source_mg\|counters" | cut -c1-50,100-220 | head -20
