#!/bin/bash
# the fused radiation stage (qk_rad_stage_fused) against the separate operators on the boxes a hierarchy produces (blocking factor 8: 8 .. 32 cells per edge, not cubes)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r6/fr && cd gpurun_out/r6/fr && R=../../..
cp $R/tests/golden/dust_shell_initial_conditions.txt initial_conditions.txt

for f in 0 1; do
  QK_AMR_VERBOSE=1 $R/quokka_amd/host/bin/ref_RadhydroShell $R/quokka_amd/host/decks/radhydro_shell_amr.in amr.n_cell="32 32 32" amr.max_level=1 amr.blocking_factor=8 amr.max_grid_size=32 max_timesteps=4 qk.fused_radiation=$f qk.dump_state=s$f.bin > log$f.txt 2>&1
  grep -c "makeLevel" log$f.txt; grep "makeLevel" log$f.txt | tail -1 | cut -c1-400
done
python - <<PY
import numpy as np
a, b = np.fromfile("s0.bin"), np.fromfile("s1.bin")
print("fused vs separate radiation operators on the hierarchy: equal in every bit:", np.array_equal(a, b), "max abs diff", float(np.abs(a - b).max()), "finite", bool(np.isfinite(a).all()))
PY
rm -f *.bin
