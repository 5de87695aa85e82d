"""Run-to-run repeatability of the drivers (digests of the final state over repeated runs): a data race or a read of uninitialised
memory shows up as more than one digest.  python profiles/tools/repeatability.py [reps]"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from quokka_amd.amr_simulation import sedov_amr_problem  # noqa: E402
from quokka_amd.multifab import Context  # noqa: E402
from quokka_amd.radhydro import shell_problem  # noqa: E402
from quokka_amd.simulation import sedov_problem  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = Context(0)
here = os.path.dirname(os.path.abspath(__file__))
tab = np.loadtxt(os.path.join(here, "..", "..", "tests", "golden", "dust_shell_initial_conditions.txt"), skiprows=1)


def digest(sim):
    h = hashlib.md5()
    for v in sim.gather_valid_local():
        h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()[:8]


def sedov(n, mgs, steps):
    s = sedov_problem(ctx, n, max_grid_size=mgs)
    for _ in range(steps):
        assert s.step()
    return digest(s)


def shell(n, mgs, steps):
    s = shell_problem(ctx, n, (tab[:, 0], tab[:, 2], tab[:, 3]), max_grid_size=mgs, pow_mode=1)
    for _ in range(steps):
        assert s.step()
    return digest(s)


def amr(n, steps):
    a = sedov_amr_problem(ctx, n, 2, max_grid_size=32, blocking_factor=8)
    for _ in range(steps):
        a.step()
    h = hashlib.md5()
    for L in a.levels:
        for v in L.gather_valid_local():
            h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()[:8]


for name, fn in (("sedov 96^3, 32^3 boxes, 20 steps", lambda: sedov(96, 32, 20)), ("sedov 80^3 ragged boxes of <= 32, 20 steps", lambda: sedov(80, 32, 20)),
                 ("shell 48^3, 16^3 boxes, 3 steps", lambda: shell(48, 16, 3)), ("sedov AMR 64^3 base, 3 levels, 24 steps", lambda: amr(64, 24))):
    ds = [fn() for _ in range(reps)]
    print(f"{name}: {len(set(ds))} distinct digest(s) in {reps} runs {sorted(set(ds))}", flush=True)
