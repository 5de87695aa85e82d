"""The N > 1 path with REAL kernels: several processes (one "rank" each) share the one GPU of the test box and talk through
torch.distributed/gloo with host-staged buffers (quokka_amd/comm.py) instead of RCCL.  Everything else is the production code:
box -> rank map, ghost plans, pack / unpack / copy kernels, the early/late overlap schedule, the fused all-reduce of CFL maxima and redo
counts.  The union of the ranks' boxes must equal the single-process run bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(rank, world, port, N, mgs, nsteps, overlap, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from quokka_amd.multifab import Context
        from quokka_amd.simulation import sedov_problem
        ctx = Context(0)
        sim = sedov_problem(ctx, N, max_grid_size=mgs, rank=rank, nranks=world)
        if overlap:
            sim.min_overlap_cells = 1
        dts = []
        for _ in range(nsteps):
            assert sim.step()
            dts.append(sim.dt_)
        groups = sim.overlap_groups() if overlap else None
        q.put((rank, [(lo, hi) for lo, hi in sim.my_boxes], [v.copy() for v in sim.gather_valid_local()], dts, len(sim.ghost.peers),
               None if groups is None else (len(groups[0][1]), len(groups[1][1]))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,overlap", [(2, False), (4, True), (8, False)])
def test_ranks_sharing_one_gpu_reproduce_the_single_process_run(ctx, world, overlap):
    from quokka_amd.simulation import sedov_problem
    N, mgs, nsteps = 32, 8, 5  # 64 boxes of 8^3
    ref = sedov_problem(ctx, N, max_grid_size=mgs)
    ref_dts = []
    for _ in range(nsteps):
        assert ref.step()
        ref_dts.append(ref.dt_)
    want = np.zeros((6, N, N, N))
    for (lo, hi), v in zip(ref.my_boxes, ref.gather_valid_local()):
        want[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = v
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = free_port()
    procs = [mpctx.Process(target=worker, args=(r, world, port, N, mgs, nsteps, overlap, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    got = np.full((6, N, N, N), np.nan)
    for rank, boxes, vals, dts, npeers, groups in results:
        assert dts == ref_dts, f"rank {rank}: time steps differ"
        assert npeers >= 1
        if overlap:
            assert groups is not None and groups[0] > 0 and groups[1] > 0, f"rank {rank}: no early/late split ({groups})"
        for (lo, hi), v in zip(boxes, vals):
            got[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = v
    assert not np.isnan(got).any(), "some box is owned by no rank"
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"
