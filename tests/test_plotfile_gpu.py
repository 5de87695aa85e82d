"""On-disk formats from the Python drivers: the plotfile holds the device state, a checkpoint restores a run
bit for bit (the reference's checkpoint_restart_test.sh criterion, on the unigrid driver)."""
import os

import numpy as np
import pytest

from quokka_amd import plotfile
from quokka_amd.simulation import sedov_problem

pytestmark = pytest.mark.gpu


def test_checkpoint_restart_is_bit_exact_and_plotfile_holds_the_state(ctx, tmp_path):
    N, mgs = 32, 16
    a = sedov_problem(ctx, N, max_grid_size=mgs)
    for _ in range(5):
        assert a.step()
    chk = str(tmp_path / "chk00005")
    plotfile.WriteCheckpointFile(a, chk)
    assert os.readlink(tmp_path / "last_chk") == "chk00005"
    for _ in range(5):
        assert a.step()
    plotfile.WritePlotFile(a, str(tmp_path / "plt00010"))
    p = plotfile.read_plotfile(str(tmp_path / "plt00010"))
    assert p.level_steps == [10] and p.time == a.tNew_ and p.levels[0].nghost == 0
    for b in range(a.lev.nboxes):
        assert np.array_equal(p.levels[0].fabs[b], a.state_new_cc_.valid(b).cpu().numpy())

    # restart on a different BoxArray (one 32^3 box instead of eight 16^3): ParallelCopy semantics of the level-0 read
    b = sedov_problem(ctx, N, max_grid_size=N)
    h = plotfile.ReadCheckpointLevel0(b, str(tmp_path / "last_chk"))
    assert h.istep == [5] and b.istep == 5
    for _ in range(5):
        assert b.step()
    assert b.tNew_ == a.tNew_ and b.istep == 10
    plotfile.WritePlotFile(b, str(tmp_path / "plt00010"))  # the first one is kept as plt00010.old.*
    old = [d for d in os.listdir(tmp_path) if d.startswith("plt00010.old.")]
    assert len(old) == 1
    q = plotfile.read_plotfile(str(tmp_path / "plt00010"))
    whole = np.zeros((6, N, N, N))
    for (lo, hi), fab in zip(p.levels[0].boxes, p.levels[0].fabs):
        whole[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = fab
    assert np.array_equal(q.levels[0].fabs[0], whole)
