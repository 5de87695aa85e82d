"""Per-launch times of the multigroup kernels on a box-filling problem: RadhydroShockMultigroup (5 photon groups, PPL fixed-slope opacities,
beta_order 1) extruded to nx x nyz x nyz cells in 128^3 boxes.  Prints one JSON line: ms per launch, ns per cell and — for the matter-radiation
exchange kernel — Newton iterations per solve.     gpurun -- 'python profiles/tools/mg_kernel_time.py [nx nyz steps]'"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

sys.path.insert(0, ROOT)
from bench import read_profile  # noqa: E402
from quokka_amd.multifab import Context  # noqa: E402
from quokka_amd.radhydro_multigroup import radshock_mg_problem  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nyz = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ctx = Context(0)
sim = radshock_mg_problem(ctx, nx, three_d=True, max_grid_size=[128, 128, 128], nyz=nyz)
for _ in range(2):
    assert sim.step()
L = ctx.L
L.qk_profile_reset(ctx.h)
L.qk_profile_enable(ctx.h, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    assert sim.step()
torch.cuda.synchronize()
el = time.perf_counter() - t0
L.qk_profile_enable(ctx.h, 0)
k = read_profile(ctx)
cells = nx * nyz * nyz
out = {"cells": cells, "groups": sim.nGroups, "ms_per_step": el / steps * 1e3, "substeps_per_step": sim.radiationCellUpdates_ / max(sim.cellUpdates_, 1),
       "newton_iterations_per_solve": sim.rad_counters["newton_iterations"] / max(sim.rad_counters["solves"], 1),
       "kernels": {n: {"launches": v[0], "ms_per_launch": v[1] / max(v[0], 1), "ns_per_cell": v[1] / max(v[0], 1) * 1e6 / cells} for n, v in sorted(k.items())}}
print(json.dumps(out))
