#!/usr/bin/env python3
"""microbenchmark of the ghost-cell interpolation of a small refined level (the config-5 geometry: one 64^3 fine box in the corner of a 128^3 coarse
box): HIP-event time per qk_InterpFromCoarse call, single time level and time-interpolated"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from quokka_amd.amr import InterpFromCoarse  # noqa: E402
from quokka_amd.multifab import Context, Level, MultiFab  # noqa: E402
from quokka_amd.simulation import Geometry  # noqa: E402

ctx = Context(0)
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 64
crse = Level(ctx, 3, [([0, 0, 0], [127, 127, 127])])
fine = Level(ctx, 3, [([0, 0, 0], [nf - 1, nf - 1, nf - 1])])
gf = Geometry(3, [256] * 3, [0.0] * 3, [1.0] * 3, [0, 0, 0])
C0 = MultiFab(crse, 6, 4)
C1 = MultiFab(crse, 6, 4)
for m in (C0, C1):
    m.storage.uniform_(1.0, 2.0)
F = MultiFab(fine, 6, 4, fill=0.0)
plan = InterpFromCoarse(crse, fine, gf, 4)
print("items", len(plan.items()), [tuple(hi[d] - lo[d] + 1 for d in range(3)) for _, _, lo, hi in plan.items()])
for name, w in (("single", (1.0, 0.0)), ("time-interpolated", (0.25, 0.75))):
    for _ in range(5):
        plan(F, C0, C1, w[0], w[1], 6, 1, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        plan(F, C0, C1, w[0], w[1], 6, 1, True)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per call; digest {float(F.storage.double().sum()):.17g}")
