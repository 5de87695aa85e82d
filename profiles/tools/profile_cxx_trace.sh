# kernel trace of the C++ host mirror on Sedov 256^3 (the unmodified reference problem file): -> gpurun_out/v4_cxx/kt.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v4_cxx; mkdir -p $O; B=$R/quokka_amd/host
rocprofv3 --kernel-trace --stats -d $O/kt -- $B/bin/ref_HydroBlast3D $B/decks/blast_unigrid_256.in max_timesteps=60 > $O/kt.log 2>&1
f=$(find $O/kt -name "*.db" | head -1); python $R/profiles/summarize_rocpd.py "$f" > $O/kt.txt 2>&1
find $O -name "*.db" -delete; rm -rf $O/kt
grep "figure-of-merit" $O/kt.log; head -22 $O/kt.txt | cut -c1-70,105-170
