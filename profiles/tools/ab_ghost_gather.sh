#!/bin/bash
# same-box A/B of the ghost fill as one gather launch (QK_GHOST_GATHER / qk.ghost_gather; qk_FillBoundary_gather) against copies + boundary rules
# (two launches; the C++ host: three, with the problem's empty setCustomBoundaryConditions): headline configuration, both hosts, interleaved
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/ab_ghost_gather.txt
: > $out
for n in 256 512; do
  for rep in 1 2 3; do
    for g in 0 1; do
      QK_GHOST_GATHER=$g python bench.py --ncell $n --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k={a:round(b,4) for a,b in d['roofline'].get('all_kernels_ms_per_launch',{}).items() if a.startswith('ghost')}
print('python host n=$n gather=$g rep=$rep value=%.1f ms_per_step=%.3f ghost kernels=%s' % (d['value'], d['ms_per_step'], json.dumps(k)))" >> $out
    done
  done
done
deck=quokka_amd/host/decks/blast_unigrid_256.in
for rep in 1 2 3; do
  for g in 0 1; do
    ./quokka_amd/host/bin/ref_HydroBlast3D $deck max_timesteps=200 qk.ghost_gather=$g hydro.rk2_carry_rhs=1 2>&1 | grep -E "figure-of-merit" | sed "s/^/cxx host (reference's HydroBlast3D file, 200 steps) gather=$g rep=$rep /" >> $out
  done
done
cat $out
