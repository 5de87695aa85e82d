#!/bin/bash
# Where does RadBeam on 4 ranks in the ancestor scheme (qk.distribute_levels=0, 8^2 level-0 boxes) first differ from one rank building the same grids?
# Per coarse-step count K: both runs, then the per-level comparison of their final plotfiles.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r6/bisect; mkdir -p $OUT/one $OUT/many; cd $OUT
R=../../../..
for K in ${STEPS:-1 2 3 4 5 6}; do
  s=$(printf "%05d" $K)
  (cd one; rm -rf plt*; QK_MAX_COARSE_STEPS=$K $R/quokka_amd/host/bin/ref_RadBeam $R/quokka_amd/host/decks/beam.in plotfile_interval=-1 checkpoint_interval=-1 amr.max_grid_size=8 \
     qk.cluster_within_parent=1 ${EXTRA} > log.txt 2>&1)
  (cd many; rm -rf plt*; QK_AMR_VERBOSE=1 QK_MAX_COARSE_STEPS=$K TMO=60 TAIL=1 $R/profiles/tools/run_cxx_ranks.sh 4 . $R/quokka_amd/host/bin/ref_RadBeam $R/quokka_amd/host/decks/beam.in \
     plotfile_interval=-1 checkpoint_interval=-1 amr.max_grid_size=8 qk.level0_distribution=interleaved qk.distribute_levels=0 ${EXTRA} > log.txt 2>&1)
  echo "=== K = $K"; grep -h "radiation source" many/rank*.log | head -3
  python ../../../profiles/tools/compare_plotfiles.py one/plt$s many/plt$s 2>&1 | grep -v "level 0 boxes" | cut -c1-260
  for a in one/afterregrid*; do [ -e $a/Header ] && [ -e many/$(basename $a)/Header ] && python ../../../profiles/tools/compare_plotfiles.py $a many/$(basename $a) 2>&1 | grep -v "level 0 boxes" | cut -c1-260; done
done
rm -rf one/plt* many/plt* one/afterregrid* many/afterregrid*
