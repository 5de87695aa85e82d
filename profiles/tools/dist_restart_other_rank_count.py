"""C++ host, distributed levels: a checkpoint written by 4 ranks after 4 coarse steps, restarted on 2 ranks and on 1 rank (all chopping their levels for 4 boxes), against
the uninterrupted 4-rank run: final plotfiles compared level by level."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))
sys.path.insert(0, ROOT)
import multirank_sweep as ms  # noqa: E402  (its run / compare helpers; the module's sweep itself runs on import only as __main__)

ARGS = ["geometry.prob_lo=0 0 0", "geometry.prob_hi=1.2 1.2 1.2", "geometry.is_periodic=0 0 0", "amr.n_cell=32 32 32", "amr.max_grid_size=8", "amr.max_level=2",
        "amr.blocking_factor=8", "amr.n_error_buf=3", "do_reflux=1", "qk.distribute_levels=1", "qk.refine_grid_layout_target=4", "qk.level0_distribution=bricks",
        "plotfile_interval=100000", "max_timesteps=8"]
with tempfile.TemporaryDirectory() as w, tempfile.TemporaryDirectory() as r2, tempfile.TemporaryDirectory() as r1:
    rc, outs, whole = ms.run("ref_HydroBlast3D", ARGS + ["checkpoint_interval=4"], w, 4, 8)
    assert all(c in (0, 1) for c in rc) and whole, outs[0][-1500:]
    chk = os.path.join(w, "chk00004")
    assert os.path.isdir(chk), os.listdir(w)
    for n, d in ((2, r2), (1, r1)):
        rc, outs, again = ms.run("ref_HydroBlast3D", ARGS + ["checkpoint_interval=-1", f"restartfile={chk}"], d, n, 8)
        assert all(c in (0, 1) for c in rc) and again, outs[0][-1500:]
        print(f"restart of the 4-rank checkpoint on {n} rank(s) against the uninterrupted 4-rank run:", ms.compare(whole, again), flush=True)
