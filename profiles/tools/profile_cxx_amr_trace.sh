# kernel trace of the C++ host mirror on BASELINE config 5 (blast_amr_maxlev2.in through the unmodified reference problem file): -> gpurun_out/<tag>/kt.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-cxx_amr}; mkdir -p $O; B=$R/quokka_amd/host
(cd $B && $B/bin/ref_HydroBlast3D $B/decks/blast_amr_maxlev2.in max_timesteps=55 hydro.rk2_carry_rhs=1 plotfile_interval=-1 checkpoint_interval=-1 > $O/plain.log 2>&1)
(cd $B && rocprofv3 --kernel-trace --stats -d $O/kt -- $B/bin/ref_HydroBlast3D $B/decks/blast_amr_maxlev2.in max_timesteps=55 hydro.rk2_carry_rhs=1 plotfile_interval=-1 checkpoint_interval=-1 > $O/kt.log 2>&1)
f=$(find $O/kt -name "*.db" | head -1); python $R/profiles/summarize_rocpd.py "$f" > $O/kt.txt 2>&1
find $O -name "*.db" -delete; rm -rf $O/kt
grep -E "figure-of-merit|elapsed|Zone-updates" $O/plain.log $O/kt.log; head -40 $O/kt.txt | cut -c1-70,105-170
