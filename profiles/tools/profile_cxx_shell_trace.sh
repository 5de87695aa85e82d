# kernel trace of the C++ host mirror on RadhydroShell 256^3 (the unmodified reference problem file, its own 50 steps): -> gpurun_out/v4_cxx_shell/kt.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v4_cxx_shell; mkdir -p $O /tmp/cs; B=$R/quokka_amd/host
cd /tmp/cs; cp $R/tests/golden/dust_shell_initial_conditions.txt initial_conditions.txt
rocprofv3 --kernel-trace --stats -d $O/kt -- $B/bin/ref_RadhydroShell $B/decks/radhydro_shell_256.in > $O/kt.log 2>&1
f=$(find $O/kt -name "*.db" | head -1); python $R/profiles/summarize_rocpd.py "$f" > $O/kt.txt 2>&1
find $O -name "*.db" -delete; rm -rf $O/kt
grep "figure-of-merit" $O/kt.log; head -24 $O/kt.txt | cut -c1-70,105-170
