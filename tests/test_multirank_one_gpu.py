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


def collect(procs, q, world, timeout=240):
    """results of all ranks; a rank that fails reports its traceback, and the others (blocked in a collective) are terminated"""
    results = []
    try:
        for _ in range(world):
            r = q.get(timeout=timeout)
            if isinstance(r, tuple) and r and r[0] == "error":
                raise AssertionError(f"rank {r[1]} failed:\n{r[2]}")
            results.append(r)
    finally:
        for p in procs:
            p.join(timeout=20)
        for p in procs:
            if p.is_alive():
                p.terminate()
    return results


def guarded(fn):
    def wrapper(rank, *args):
        q = args[-1]
        try:
            fn(rank, *args)
        except BaseException:  # noqa: BLE001 - report and let the parent tear the group down
            import traceback
            q.put(("error", rank, traceback.format_exc()))
            os._exit(1)
    wrapper.__name__ = fn.__name__
    return wrapper


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


def run_worker(rank, *args):
    guarded(worker)(rank, *args)


def run_amr_worker(rank, *args):
    guarded(amr_worker)(rank, *args)


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
    procs = [mpctx.Process(target=run_worker, args=(r, world, port, N, mgs, nsteps, overlap, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = collect(procs, q, world)
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


def amr_worker(rank, world, port, N, nsteps, q, distribution="bricks", mgs=16):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from quokka_amd.amr_simulation import sedov_amr_problem
        from quokka_amd.multifab import Context
        ctx = Context(0)
        amr = sedov_amr_problem(ctx, N, 2, max_grid_size=mgs, blocking_factor=8, rank=rank, nranks=world, level0_distribution=distribution)
        m0, e0 = amr.composite_sum(0), amr.composite_sum(4)
        for _ in range(nsteps):
            amr.step()
        m1, e1 = amr.composite_sum(0), amr.composite_sum(4)
        out = []
        for L in amr.levels:
            out.append(([(lo, hi) for lo, hi in L.all_boxes], list(L.owner), [(lo, hi) for lo, hi in L.my_boxes], [v.copy() for v in L.gather_valid_local()]))
        q.put((rank, out, amr.tNew_, (abs(m1 - m0) / m0, abs(e1 - e0) / e0), list(amr.istep)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,distribution,N,mgs", [(2, "interleaved", 32, 16), (4, "interleaved", 32, 16), (4, "bricks", 32, 16), (8, "bricks", 64, 32)])
def test_amr_hierarchy_across_ranks_matches_one_rank(ctx, world, distribution, N, mgs):
    """Sedov, max_level = 2.  32^3 base grid in 16^3 boxes over 2 / 4 ranks, and the geometry of tests/blast_amr_maxlev2.in (BASELINE config 5: 256^3
    in 128^3 boxes, blocking factor 32, 8 ranks) scaled by four: 64^3 in 32^3 boxes, blocking factor 8, 8 ranks sharing the GPU.  Every level has a
    box -> rank map of its own (space-filling curve; a level with fewer boxes than ranks is chopped, AMReX's refine_grid_layout): the parent's
    data reach the fine boxes, averaged-down and refluxed data the coarse boxes, through pack -> point-to-point -> unpack plans (CoarseShadow,
    DistFluxRegister).  Same grids and time steps as ONE rank chopping for the same box count; states agree to rounding (the two parts of a
    flux register are added to the state one after the other), mass and energy are conserved; with 8 ranks every rank owns a box of every level."""
    from quokka_amd.amr_simulation import sedov_amr_problem
    nsteps = 6
    ref = sedov_amr_problem(ctx, N, 2, max_grid_size=mgs, blocking_factor=8, refine_grid_layout_target=world)
    for _ in range(nsteps):
        ref.step()
    assert ref.finest_level == 2
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = free_port()
    procs = [mpctx.Process(target=run_amr_worker, args=(r, world, port, N, nsteps, q, distribution, mgs)) for r in range(world)]
    for p in procs:
        p.start()
    results = collect(procs, q, world)
    results.sort(key=lambda r: r[0])
    for rank, levels, tnew, drift, istep in results:
        assert tnew == ref.tNew_ and istep == ref.istep, f"rank {rank}: time stepping differs"
        assert len(levels) == 3
        assert drift[0] <= 2e-13 and drift[1] <= 2e-13, f"rank {rank}: composite mass / energy drift {drift}"
    worst = 0.0
    for l, L in enumerate(ref.levels):
        n = L.geom.n_cell
        want = np.full((6, n[2], n[1], n[0]), np.nan)
        for (lo, hi), v in zip(L.my_boxes, L.gather_valid_local()):
            want[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = v
        got = np.full_like(want, np.nan)
        owners_seen = set()
        for rank, levels, *_ in results:
            all_boxes, owner, mine, vals = levels[l]
            assert sorted(map(str, all_boxes)) == sorted(map(str, [(list(lo), list(hi)) for lo, hi in L.all_boxes])), f"level {l}: grids differ on rank {rank}"
            assert [b for b, o in zip(all_boxes, owner) if o == rank] == mine
            owners_seen.update(owner)
            if world == 8:
                assert mine, f"rank {rank} owns no box of level {l} ({len(all_boxes)} boxes)"
            for (lo, hi), v in zip(mine, vals):
                got[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = v
        assert np.array_equal(np.isnan(got), np.isnan(want)), f"level {l}: coverage differs"
        m = ~np.isnan(want)
        scale = np.abs(want[m]).max()
        worst = max(worst, float(np.abs(got[m] - want[m]).max() / scale))
        if l == 0 or world == 8:
            assert len(owners_seen) == world
        elif l > 0:
            assert len(owners_seen) > 1, f"level {l} lives on one rank"
    assert worst <= 1e-13, worst


def rad_amr_worker(rank, world, port, N, nsteps, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from quokka_amd.amr_simulation import rad_pulse_amr_problem
        from quokka_amd.multifab import Context
        ctx = Context(0)
        amr = rad_pulse_amr_problem(ctx, N, 1, max_grid_size=16, blocking_factor=8, rank=rank, nranks=world, tag_threshold=1.02)
        e0 = amr.composite_sum(5) + amr.composite_sum(6)
        for _ in range(nsteps):
            amr.step()
        e1 = amr.composite_sum(5) + amr.composite_sum(6)
        out = []
        for L in amr.levels:
            out.append(([(lo, hi) for lo, hi in L.all_boxes], list(L.owner), [(lo, hi) for lo, hi in L.my_boxes], [v.copy() for v in L.gather_valid_local()]))
        q.put((rank, out, amr.tNew_, abs(e1 - e0) / e0, list(amr.istep)))
    finally:
        dist.destroy_process_group()


def run_rad_amr_worker(rank, *args):
    guarded(rad_amr_worker)(rank, *args)


@pytest.mark.parametrize("world", [2, 4])
def test_radiation_amr_hierarchy_across_ranks_matches_one_rank(ctx, world):
    """The radiation pulse on a dynamically refined hierarchy (32^3 base grid in 16^3 boxes over 2 / 4 ranks), the refined level with a box -> rank
    map of its own: the fine part of the radiation block's flux register travels to the coarse owners like the hydro block's.  Same grids and
    time steps as one rank, states to rounding, E_int + E_rad of the composite grid at the Newton tolerance."""
    from quokka_amd.amr_simulation import rad_pulse_amr_problem
    N, nsteps = 32, 8
    ref = rad_pulse_amr_problem(ctx, N, 1, max_grid_size=16, blocking_factor=8, tag_threshold=1.02, refine_grid_layout_target=world)
    for _ in range(nsteps):
        ref.step()
    assert ref.finest_level == 1
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = free_port()
    procs = [mpctx.Process(target=run_rad_amr_worker, args=(r, world, port, N, nsteps, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = collect(procs, q, world)
    results.sort(key=lambda r: r[0])
    worst = 0.0
    for rank, levels, tnew, drift, istep in results:
        assert tnew == ref.tNew_ and istep == ref.istep, f"rank {rank}: time stepping differs"
        assert len(levels) == 2 and drift <= 1e-11, (rank, drift)
    for l, L in enumerate(ref.levels):
        n = L.geom.n_cell
        want = np.full((10, n[2], n[1], n[0]), np.nan)
        for (lo, hi), v in zip(L.my_boxes, L.gather_valid_local()):
            want[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = v
        got = np.full_like(want, np.nan)
        for rank, levels, *_ in results:
            all_boxes, owner, mine, vals = levels[l]
            assert sorted(map(str, all_boxes)) == sorted(map(str, [(list(lo), list(hi)) for lo, hi in L.all_boxes])), f"level {l}: grids differ on rank {rank}"
            for (lo, hi), v in zip(mine, vals):
                got[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = v
        assert np.array_equal(np.isnan(got), np.isnan(want)), f"level {l}: coverage differs"
        m = ~np.isnan(want)
        for c in (0, 4, 5, 6):
            mc = m[c]
            worst = max(worst, float(np.abs(got[c][mc] - want[c][mc]).max() / np.abs(want[c][mc]).max()))
    assert worst <= 1e-13, worst


@pytest.mark.parametrize("world", [2, 4])
def test_bench_selftest_passes_on_ranks_sharing_the_gpu(world):
    """`python bench.py --gpus N --selftest` — the command to run first on a multi-GPU lease — through the real launcher
    (torch.distributed.run), with the ranks sharing this GPU over gloo (QK_BENCH_ONE_GPU_TEST; on N GPUs the same command goes over RCCL):
    per-box digests of the N-rank run == the one-rank run, dt equal on all ranks, early / late schedule active."""
    import json
    import subprocess
    env = dict(os.environ, QK_BENCH_ONE_GPU_TEST="1", OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--selftest"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{") and "selftest" in l]
    assert p.returncode == 0 and len(lines) == 1, p.stdout[-1500:] + p.stderr[-1500:]
    v = json.loads(lines[0])
    assert v["selftest"] == "PASS" and v["n_gpus"] == world and v["boxes_differing_from_one_rank"] == 0 and v["dt_equal_on_all_ranks"], v
    assert v["boxes"] == 8 * world and v["early_late_boxes_rank0"] is not None, v


def test_bench_multi_gpu_line_dry_run_with_eight_ranks():
    """`python bench.py --gpus 8` as the driver launches it on an 8-GPU node, DRY: the eight ranks share this GPU over gloo on a small blast
    (QK_BENCH_ONE_GPU_TEST; marked `dry_run` in the line).  The complete line shape: the self-test verdict first (N ranks == one rank, bit for
    bit), the ghost_exchange block (bytes per fill, exposed milliseconds, peers), config 5 (amr_maxlev2, strong scaling) on the same ranks."""
    import json
    import subprocess
    world = 8
    env = dict(os.environ, QK_BENCH_ONE_GPU_TEST="1", OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert p.returncode == 0 and len(lines) == 1, p.stdout[-1500:] + p.stderr[-2500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 3 and d["scaling"] == "weak" and "dry_run" in d
    assert d["selftest"]["selftest"] == "PASS" and d["selftest"]["boxes_differing_from_one_rank"] == 0, d["selftest"]
    g = d["config"]["ghost_exchange"]
    assert g["peers_rank0"] >= 3 and g["bytes_sent_per_fill_rank0"] > 0 and g["fills_timed"] == 2 * 3 and g["exposed_ms_per_fill_max_over_ranks"] >= 0.0, g
    assert sum(g["exposed_ms_histogram_rank0"]["fills"]) == g["fills_timed"]
    a = d["amr_maxlev2"]
    assert a["scaling"] == "strong" and a["value"] > 0 and len(a["config"]["cells_per_level"]) == 3, a
