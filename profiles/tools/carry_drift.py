"""Growth of the difference between the two forms of the RK2 average (exact: flux_rk2 face by face; carried half step) over a long run of
the Sedov blast, and — as the yardstick — between two EXACT-form runs whose initial energy differs by one unit in the last place in one cell.
usage: python profiles/tools/carry_drift.py [ncell] [nsteps]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from quokka_amd.multifab import Context
from quokka_amd.simulation import sedov_problem

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
ctx = Context(0)
a, b, c = (sedov_problem(ctx, N, max_grid_size=128) for _ in range(3))
b.rk2_carry_rhs = True
# c: exact form, the energy of the blast cell (0, 0, 0) one unit in the last place up (relative size 2e-16: what ONE rounding difference is)
v = c.state_new_cc_.valid(0)
e = v[4, 0, 0, 0].item()
v[4, 0, 0, 0] = np.nextafter(e, 2 * e)
c.fillBoundaryConditions(c.state_new_cc_)


def rel_l1(x, y):
    worst = 0.0
    for n in range(6):
        num = den = 0.0
        for k in range(x.lev.nboxes):
            p, q = x.state_new_cc_.valid(k)[n], y.state_new_cc_.valid(k)[n]
            num += float((p - q).abs().sum(dtype=torch.float64))
            den += float(p.abs().sum(dtype=torch.float64))
        worst = max(worst, num / max(den, 1e-300))
    return worst


marks = [1, 10, 30, 100, 300, 1000, 2000, 3000, 4000, 6000, 8000, 10000, 12000]
for it in range(1, nsteps + 1):
    assert a.step() and b.step() and c.step()
    if it in marks or it == nsteps:
        print(f"step {it:6d}  t = {a.tNew_:.4e}  dt exact/carry/perturbed = {a.dt_:.6e} {b.dt_:.6e} {c.dt_:.6e}   "
              f"rel L1: carry vs exact {rel_l1(a, b):.2e}   exact(blast energy + 1 ulp) vs exact {rel_l1(a, c):.2e}", flush=True)
