# The same digest with and without the pooled arena / the single hardware queue, over a run long enough for the grids to change (300 coarse steps):
# bash profiles/tools/cxx_amr_repeat_env.sh
R=$GRAFT_REPO_ROOT; B=$R/quokka_amd/host; W=/tmp/cxx_amr_repeat_env; rm -rf $W; mkdir -p $W
n=0
for envs in "QK_NONE=1" "QK_NONE=1" "QK_DEVICE_ARENA=0" "QK_OWN_COMPUTE_STREAM=1" "QK_DEVICE_ARENA=0 QK_OWN_COMPUTE_STREAM=1"; do
  n=$((n+1)); mkdir -p $W/$n; cd $W/$n
  env $envs $B/bin/ref_HydroBlast3D $B/decks/blast_amr_maxlev2.in max_timesteps=300 hydro.rk2_carry_rhs=1 plotfile_interval=100000 checkpoint_interval=-1 > log 2>&1
  d=$(ls -d plt* 2>/dev/null | tail -1)
  (cd $d && find . -name "Cell_D_*" | sort | xargs cat | md5sum | cut -c1-12) > digest
  echo "$envs: $(cat digest)  $(grep -o '\[[0-9.]* Mupdates/s\]' log)  grids: $(grep -o '([0-9]* grids)' log | tr '\n' ' ')"
done
echo "distinct digests: $(cat $W/*/digest | sort -u | wc -l)"
