"""C++ host, distributed levels, 4 ranks: every array allocated with NaN bit patterns (QK_POISON=1) against the ordinary zero-filled allocation — a read of a shadow cell
that no copy, boundary condition or interpolation wrote (a hole of the ParallelCopy plan, a ring cell outside the parent level) would show as NaN or as a difference."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))
sys.path.insert(0, ROOT)
import tempfile  # noqa: E402

import multirank_sweep as ms  # noqa: E402

for name, exe, deck, over, nranks, steps in ms.CASES:
    if name not in ("blast2d", "beam8", "shocktube_cma", "periodic3d", "four_levels", "shadow"):
        continue
    base = ([os.path.join(ms.HOST, "decks", deck)] if deck else []) + ["plotfile_interval=100000", "checkpoint_interval=-1", "qk.distribute_levels=1"] + over
    with tempfile.TemporaryDirectory() as d0, tempfile.TemporaryDirectory() as d1:
        os.environ.pop("QK_POISON", None)
        rc0, o0, p0 = ms.run(exe, base, d0, nranks, steps)
        os.environ["QK_POISON"] = "1"
        rc1, o1, p1 = ms.run(exe, base, d1, nranks, steps)
        os.environ.pop("QK_POISON", None)
        if not (all(r in (0, 1) for r in rc0 + rc1) and p0 and p1):
            print(f"{name:14s} FAILED rc {rc0} {rc1}: {[ln for ln in (o1[0] + o0[0]).splitlines() if 'Abort' in ln][:2]}", flush=True)
            continue
        print(f"{name:14s} poisoned against zero-filled allocations, {nranks} ranks: {ms.compare(p0, p1)}", flush=True)
