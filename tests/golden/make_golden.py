"""Generates the committed golden fixtures from the CPU oracle (run from the repo root).
ppm1d_sod_exact.txt is the reference's own data table extern/ppm1d/output (exact Sod solution at t = 0.4)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.pyoracle import SEDOV, SOD, Oracle  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
o = Oracle("direct")
s = o.sim(SOD, 1, [1024], [0, 0, 0], [5, 1, 1], [0, 1, 1])
assert s.evolve()
np.save(os.path.join(here, "sod_1024_final.npy"), s.valid()[:, 0, 0, :])
s = o.sim(SEDOV, 3, [32] * 3, [0, 0, 0], [1.2] * 3, [0, 0, 0])
for _ in range(10):
    assert s.step()
np.save(os.path.join(here, "sedov_32_step10.npy"), s.valid(0))
# ghost-filled Sedov state with a developed shock: input vector for the per-operator GPU parity tests
np.save(os.path.join(here, "sedov_32_step10_ghosted.npy"), (s.fill_ghosts(0, s.time), s.state(0))[1])
