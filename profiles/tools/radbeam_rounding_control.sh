#!/bin/bash
# Control experiment for tests/test_host_mirror_gpu.py::test_distributed_levels_of_other_solvers_in_the_cxx_host: how fast does RadBeam amplify a rounding-level
# difference in the reflux additions?  One rank against four ranks in the ROUND-5 multi-rank mode (levels clustered inside the level-0 boxes, reflux
# increments folded with SumBoundary: additions reassociated), level-0 state compared after N coarse steps.
cd "$(dirname "$0")/../.." || exit 1
OUT=${OUT:-gpurun_out/r6/ctl}; mkdir -p $OUT; cd $OUT
R=../../..
for st in 8 10 12 16 20; do
  QK_MAX_COARSE_STEPS=$st $R/quokka_amd/host/bin/ref_RadBeam $R/quokka_amd/host/decks/beam.in plotfile_interval=-1 checkpoint_interval=-1 amr.max_grid_size=${MGS:-32} \
     qk.cluster_within_parent=1 qk.dump_state=one.bin > one.log 2>&1
  QK_MAX_COARSE_STEPS=$st TMO=60 TAIL=1 $R/profiles/tools/run_cxx_ranks.sh 4 . $R/quokka_amd/host/bin/ref_RadBeam $R/quokka_amd/host/decks/beam.in plotfile_interval=-1 \
     checkpoint_interval=-1 amr.max_grid_size=${MGS:-32} qk.level0_distribution=${DIST:-bricks} qk.dump_state=many.bin > many.log 2>&1
  python - <<PY
import sys, numpy as np
sys.path.insert(0, "$R")
from quokka_amd.simulation import chop_domain, distribute_boxes, distribute_boxes_interleaved
M = ${MGS:-32}; boxes = chop_domain([128, 128, 1], [M, M, M]); owner = (distribute_boxes_interleaved if "${DIST:-bricks}" == "interleaved" else distribute_boxes)(boxes, 4, [128, 128, 1], [M, M, M])
one = np.fromfile("one.bin").reshape(len(boxes), -1, M * M)
parts = [np.fromfile(f"many.bin.rank{r}") for r in range(4)]
n = one.shape[1] * M * M; cur = [0] * 4; many = []
for r in owner:
    many.append(parts[r][cur[r]:cur[r] + n]); cur[r] += n
many = np.concatenate(many).reshape(one.shape)
print($st, [float(np.abs(one[:, c] - many[:, c]).max() / np.abs(one[:, c]).max()) for c in (6, 7, 8)])
PY
done
rm -rf *.bin* plt* chk*
