"""throughput of a static 2-level hierarchy with many small fine boxes: python profiles/tools/amr_static.py [base N] [fine box size]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

from quokka_amd.amr_simulation import sedov_amr_problem
from quokka_amd.multifab import Context
from quokka_amd.simulation import chop_domain

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ctx = Context(0)
# refine the corner octant: fine cells [0, N)^3 (= coarse [0, N/2)^3)
fine = chop_domain([N, N, N], [bs] * 3)
amr = sedov_amr_problem(ctx, N, 1, max_grid_size=128, blocking_factor=bs, static_fine_boxes=[fine])
print(f"levels {amr.finest_level + 1}; boxes {[L.lev.nboxes for L in amr.levels]}; cells {[amr.CountCells(l) for l in range(amr.finest_level + 1)]}", flush=True)
for _ in range(3):
    amr.step()
torch.cuda.synchronize()
u0, t0 = amr.cellUpdates_, time.perf_counter()
n = 20
for _ in range(n):
    amr.step()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"{n} coarse steps in {el:.3f} s ({el / n * 1e3:.2f} ms each): {(amr.cellUpdates_ - u0) / el / 1e6:.1f} Mcell-updates/s")
