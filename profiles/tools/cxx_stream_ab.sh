R=$GRAFT_REPO_ROOT; B=$R/quokka_amd/host; cd $B
ARGS="$B/decks/blast_amr_maxlev2.in max_timesteps=55 hydro.rk2_carry_rhs=1 plotfile_interval=-1 checkpoint_interval=-1"
for rep in 1 2 3; do for a in 0 1; do echo "own_stream=$a $(QK_OWN_COMPUTE_STREAM=$a $B/bin/ref_HydroBlast3D $ARGS 2>&1 | grep figure-of-merit)"; done; done
for rep in 1 2; do for a in 0 1; do echo "unigrid own_stream=$a $(QK_OWN_COMPUTE_STREAM=$a $B/bin/ref_HydroBlast3D $B/decks/blast_unigrid_256.in max_timesteps=200 hydro.rk2_carry_rhs=1 plotfile_interval=-1 checkpoint_interval=-1 2>&1 | grep figure-of-merit)"; done; done
