# HIP API time of the C++ host on BASELINE config 5 (which runtime calls the host thread waits in): -> gpurun_out/<tag>/hip_api_stats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-cxx_amr_hipapi}; mkdir -p $O; B=$R/quokka_amd/host
(cd $B && rocprofv3 --hip-runtime-trace --stats --output-format csv -d $O/ht -- $B/bin/ref_HydroBlast3D $B/decks/blast_amr_maxlev2.in max_timesteps=55 hydro.rk2_carry_rhs=1 plotfile_interval=-1 checkpoint_interval=-1 > $O/ht.log 2>&1)
f=$(find $O/ht -name "*hip_api_stats.csv" | head -1)
[ -n "$f" ] && head -25 "$f" > $O/hip_api_stats.txt
find $O/ht -type f ! -name "*stats*" -delete
grep figure-of-merit $O/ht.log; cat $O/hip_api_stats.txt
