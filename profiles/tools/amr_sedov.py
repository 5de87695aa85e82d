"""Sedov AMR (BASELINE config 5 geometry: 256^3 base, max_level 2, blocking_factor 32, max_grid_size 128) on one GPU:
python profiles/tools/amr_sedov.py [nsteps] [base N]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

from quokka_amd.amr_simulation import sedov_amr_problem
from quokka_amd.multifab import Context

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ctx = Context(0)
t0 = time.time()
amr = sedov_amr_problem(ctx, N, 2, max_grid_size=128, blocking_factor=32)
print(f"init {time.time() - t0:.1f} s; levels {amr.finest_level + 1}; boxes {[L.lev.nboxes for L in amr.levels]}; cells {[amr.CountCells(l) for l in range(amr.finest_level + 1)]}", flush=True)
m0, e0 = amr.composite_sum(0), amr.composite_sum(4)
for _ in range(3):
    amr.step()
torch.cuda.synchronize()
u0, t0 = amr.cellUpdates_, time.perf_counter()
for it in range(nsteps):
    amr.step()
    if (it + 1) % 50 == 0:
        torch.cuda.synchronize()
        print(f"  step {it + 1}: t = {amr.tNew_:.3e}, boxes {[L.lev.nboxes for L in amr.levels]}, cells {[amr.CountCells(l) for l in range(amr.finest_level + 1)]}, "
              f"{(amr.cellUpdates_ - u0) / (time.perf_counter() - t0) / 1e6:.1f} Mcell-updates/s so far", flush=True)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"{nsteps} coarse steps in {el:.2f} s: {(amr.cellUpdates_ - u0) / el / 1e6:.1f} Mcell-updates/s; per level {amr.cellUpdatesEachLevel_}; "
      f"boxes {[L.lev.nboxes for L in amr.levels]}; t = {amr.tNew_:.4e}")
m1, e1 = amr.composite_sum(0), amr.composite_sum(4)
print(f"mass drift {abs(m1 - m0) / m0:.2e}, energy drift {abs(e1 - e0) / e0:.2e}")
