# kernel trace of the C++ host mirror on RadhydroShell 256^3 (the unmodified reference problem file, source re-evaluated every call): -> gpurun_out/r4_cxxshell/kt.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_cxxshell; mkdir -p $O; B=$R/quokka_amd/host
mkdir -p /tmp/shellrun && cp $R/tests/golden/dust_shell_initial_conditions.txt /tmp/shellrun/initial_conditions.txt && cd /tmp/shellrun   # the problem opens ./initial_conditions.txt
rocprofv3 --kernel-trace --stats -d $O/kt -- $B/bin/ref_RadhydroShell $B/decks/radhydro_shell_256.in max_timesteps=12 plotfile_interval=-1 checkpoint_interval=-1 hydro.rk2_carry_rhs=1 radiation.source_is_time_independent=${SRC_ONCE:-0} > $O/kt.log 2>&1
cd $R
f=$(find $O/kt -name "*.db" | head -1); python $R/profiles/summarize_rocpd.py "$f" > $O/kt.txt 2>&1
find $O -name "*.db" -delete; rm -rf $O/kt
grep "figure-of-merit\|qk counters" $O/kt.log; head -26 $O/kt.txt | cut -c1-90,118-175
