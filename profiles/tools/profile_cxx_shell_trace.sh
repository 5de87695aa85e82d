# rocprofv3 kernel statistics of the reference's unchanged RadhydroShell file on the C++ host (deck of BASELINE config 4, 8 steps) -> gpurun_out/<tag>/kt.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-cxx_shell_kt}; mkdir -p $O
tmp=$(mktemp -d); cp $R/tests/golden/dust_shell_initial_conditions.txt $tmp/initial_conditions.txt; cd $tmp
rocprofv3 --kernel-trace --stats -d $O/kt -- $R/quokka_amd/host/bin/ref_RadhydroShell $R/quokka_amd/host/decks/radhydro_shell_256.in max_timesteps=8 plotfile_interval=-1 checkpoint_interval=-1 hydro.rk2_carry_rhs=1 > $O/kt.log 2>&1
cd $R
f=$(find $O/kt -name "*.db" | head -1); python profiles/summarize_rocpd.py "$f" > $O/kt.txt 2>&1
rm -rf $O/kt $tmp
grep -E "figure-of-merit" $O/kt.log; head -30 $O/kt.txt | cut -c1-70,105-175
