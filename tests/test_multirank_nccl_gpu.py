"""The N > 1 path over RCCL itself: one process per GPU, backend "nccl", device tensors through torch.distributed P2P (what bench.py --gpus N
runs).  Needs at least two visible GPUs (RCCL refuses two ranks on one device); on a one-GPU box the test is skipped and the same code is
covered by tests/test_multirank_one_gpu.py (gloo, host-staged).  The union of the ranks' boxes must equal the single-process run bit for bit,
with the early / late overlap schedule enabled."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def nccl_worker(rank, world, port, N, mgs, nsteps, q):
    import torch.distributed as dist
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        from quokka_amd.multifab import Context
        from quokka_amd.simulation import sedov_problem
        ctx = Context(rank)
        sim = sedov_problem(ctx, N, max_grid_size=mgs, rank=rank, nranks=world)
        sim.min_overlap_cells = 1
        dts = []
        for _ in range(nsteps):
            assert sim.step()
            dts.append(sim.dt_)
        groups = sim.overlap_groups()
        q.put((rank, [(lo, hi) for lo, hi in sim.my_boxes], [v.copy() for v in sim.gather_valid_local()], dts, len(sim.ghost.peers),
               None if groups is None else (len(groups[0][1]), len(groups[1][1])), dist.get_backend()))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException:  # noqa: BLE001
        import traceback
        q.put(("error", rank, traceback.format_exc()))
        os._exit(1)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_ranks_over_rccl_reproduce_the_single_process_run(ctx, world):
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, {torch.cuda.device_count()} visible")
    from test_multirank_one_gpu import collect, free_port
    from quokka_amd.simulation import sedov_problem
    N, mgs, nsteps = 64, 16, 6  # 64 boxes of 16^3
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
    procs = [mpctx.Process(target=nccl_worker, args=(r, world, port, N, mgs, nsteps, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = collect(procs, q, world)
    got = np.full((6, N, N, N), np.nan)
    for rank, boxes, vals, dts, npeers, groups, backend in results:
        assert backend == "nccl"
        assert dts == ref_dts, f"rank {rank}: time steps differ"
        assert npeers >= 1
        assert groups is not None and groups[0] > 0 and groups[1] > 0, f"rank {rank}: no early/late split ({groups})"
        for (lo, hi), v in zip(boxes, vals):
            got[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = v
    assert not np.isnan(got).any(), "some box is owned by no rank"
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"
