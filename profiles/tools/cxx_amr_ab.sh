# C++ host on BASELINE config 5: figure of merit with the device arena on / off, and the host-phase profile: bash profiles/tools/cxx_amr_ab.sh TAG
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-cxx_amr_ab}; mkdir -p $O; B=$R/quokka_amd/host; cd $B
ARGS="$B/decks/blast_amr_maxlev2.in max_timesteps=55 hydro.rk2_carry_rhs=1 plotfile_interval=-1 checkpoint_interval=-1"
for rep in 1 2; do
  for a in 1 0; do
    QK_DEVICE_ARENA=$a $B/bin/ref_HydroBlast3D $ARGS > $O/arena${a}_$rep.log 2>&1
    echo "arena=$a rep=$rep $(grep figure-of-merit $O/arena${a}_$rep.log)"
  done
done
QK_AMR_HOSTPROF=1 $B/bin/ref_HydroBlast3D $ARGS > $O/hostprof.log 2>&1
grep -E "figure-of-merit|host phase|elapsed" $O/hostprof.log
