# Run-to-run repeatability of the C++ host on the config-5 hierarchy (the reference's unchanged HydroBlast3D file, blast_amr_maxlev2.in, 40 coarse steps):
# md5 of every level's data files of the final plotfile over REPS runs — a race, a stale pooled block or a missing stream dependency shows as a second digest.
# usage: bash profiles/tools/cxx_amr_repeat.sh [REPS] [extra deck arguments]
reps=${1:-8}; shift
R=$GRAFT_REPO_ROOT; B=$R/quokka_amd/host; W=/tmp/cxx_amr_repeat; rm -rf $W; mkdir -p $W
for r in $(seq 1 $reps); do
  mkdir -p $W/$r; cd $W/$r
  $B/bin/ref_HydroBlast3D $B/decks/blast_amr_maxlev2.in max_timesteps=40 hydro.rk2_carry_rhs=1 plotfile_interval=100000 checkpoint_interval=-1 "$@" > log 2>&1
  d=$(ls -d plt* 2>/dev/null | tail -1)
  if [ -z "$d" ]; then echo "run $r: no plotfile"; tail -3 log; continue; fi
  (cd $d && find . -name "Cell_D_*" | sort | xargs cat | md5sum | cut -c1-12) > digest
  echo "run $r: $(cat digest)  $(grep -o '\[[0-9.]* Mupdates/s\]' log)"
done
echo "distinct digests: $(cat $W/*/digest | sort -u | wc -l) in $reps runs"
