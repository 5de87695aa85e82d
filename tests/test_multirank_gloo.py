"""world_size-2 (and 4) CPU test of the N > 1 path: box -> rank map, ghost-exchange plan (host logic of the C-ABI),
wire order and the torch.distributed exchange protocol of GhostExchange.fill_with, plus the scalar all-reduces of the
driver.  The bytes inside a rank are moved by numpy here (the GPU box moves them with the pack/copy/unpack kernels that
execute the very same plan items); the result must equal the oracle's single-process ghost fill, cell for cell."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def region(fab, begin, lo, hi, shift=(0, 0, 0)):
    """numpy view of fab[(ncomp), z, y, x] over the index region [lo, hi] shifted by -shift"""
    sl = [slice(None)]
    for d in (2, 1, 0):
        sl.append(slice(lo[d] - shift[d] - begin[d], hi[d] - shift[d] - begin[d] + 1))
    return fab[tuple(sl)]


def worker(rank, world, port, problem, N, mgs, periodic, overlapped, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.pyoracle import SEDOV, Oracle
        from quokka_amd import capi
        from quokka_amd.multifab import Level, PlanningContext
        from quokka_amd.simulation import GhostExchange, Geometry, chop_domain, distribute_boxes

        ng, nc = 4, 6
        # reference solution: the oracle fills ALL boxes in one process
        o = Oracle("direct")
        so = o.sim(problem, 3, [N] * 3, [0, 0, 0], [1.2] * 3, periodic, max_grid_size=[mgs] * 3)
        for _ in range(2):
            assert so.step()
        unfilled = [so.state(b).copy() for b in range(so.nboxes)]
        so.fill_ghosts(0, so.time)
        filled = [so.state(b) for b in range(so.nboxes)]

        geom = Geometry(3, [N] * 3, [0.0] * 3, [1.2] * 3, list(periodic))
        all_boxes = chop_domain(geom.n_cell, [mgs] * 3)
        owner = distribute_boxes(all_boxes, world, geom.n_cell, [mgs] * 3)
        assert sorted(set(owner)) == list(range(world)), owner
        mine = [g for g, r in enumerate(owner) if r == rank]
        ctx = PlanningContext()
        lev = Level(ctx, 3, [all_boxes[g] for g in mine])
        bcs = []
        for c in range(nc):
            lo = [capi.BC_REFLECT_ODD if c == 1 + d else capi.BC_REFLECT_EVEN for d in range(3)]
            bcs.append((lo, list(lo)))
        ex = GhostExchange(lev, geom, nc, ng, all_boxes, owner, rank, bcs)
        assert len(ex.peers) == world - 1 or N // mgs > 2

        fabs = [unfilled[g].copy() for g in mine]  # (nc, z, y, x) with ghosts
        begins = [[all_boxes[g][0][d] - ng for d in range(3)] for g in mine]

        def pack(k, sbuf):
            buf = sbuf.numpy()
            for db, sb, lo, hi, sh, off in ex.items(1, k):
                r = region(fabs[sb], begins[sb], lo, hi, sh)
                buf[off:off + r.size] = r.reshape(-1)

        def local():
            for db, sb, lo, hi, sh, off in ex.items(0):
                region(fabs[db], begins[db], lo, hi)[...] = region(fabs[sb], begins[sb], lo, hi, sh)

        def unpack(k, rbuf):
            buf = rbuf.numpy()
            for db, sb, lo, hi, sh, off in ex.items(2, k):
                r = region(fabs[db], begins[db], lo, hi)
                r[...] = buf[off:off + r.size].reshape(r.shape)

        late = set(ex.remote_boxes())

        def physbc(which):
            dom_hi = N - 1
            for db, sb, lo, hi, sh, off in ex.items(3):
                if (which == capi.BOXES_LOCAL_ONLY and db in late) or (which == capi.BOXES_REMOTE_DEPENDENT and db not in late):
                    continue
                f = fabs[db]
                for k in range(lo[2], hi[2] + 1):
                    for j in range(lo[1], hi[1] + 1):
                        for i in range(lo[0], hi[0] + 1):
                            idx, src, sign = (i, j, k), [i, j, k], np.ones(nc)
                            for d in range(3):
                                if idx[d] < 0:
                                    src[d] = -idx[d] - 1
                                    sign[1 + d] *= -1.0
                                elif idx[d] > dom_hi:
                                    src[d] = 2 * dom_hi - idx[d] + 1
                                    sign[1 + d] *= -1.0
                            b0 = begins[db]
                            f[:, k - b0[2], j - b0[1], i - b0[0]] = sign * f[:, src[2] - b0[2], src[1] - b0[1], src[0] - b0[0]]

        # overlapped protocol: the boxes that need nothing from other ranks must be complete while the wire is busy
        early_ok = []

        def between():
            early_ok.extend(np.array_equal(fabs[n], filled[g]) for n, g in enumerate(mine) if n not in late)

        ex.fill_with(pack, local, unpack, physbc, between if overlapped else None)
        ok = all(np.array_equal(fabs[n], filled[g]) for n, g in enumerate(mine)) and all(early_ok)
        if overlapped:
            ok = ok and len(early_ok) == len(mine) - len(late)
        nbad = sum(int((fabs[n] != filled[g]).sum()) for n, g in enumerate(mine))
        n_early = len(mine) - len(late)

        # scalar collectives of the driver (dt / CFL max, FOFC redo count)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        c = torch.tensor([rank + 10], dtype=torch.int64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        ok = ok and float(t.item()) == float(world) and int(c.item()) == sum(r + 10 for r in range(world))
        q.put((rank, ok, nbad, len(ex.peers), n_early))
    finally:
        dist.destroy_process_group()


def run(world, problem, N, mgs, periodic, overlapped=False):
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = free_port()
    procs = [mpctx.Process(target=worker, args=(r, world, port, problem, N, mgs, periodic, overlapped, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(results)


@pytest.mark.parametrize("world", [2, 4])
def test_ghost_exchange_reflecting_octant(world):
    """Sedov octant (reflecting walls), 16^3 in 8^3 boxes: 8 boxes over 2 / 4 ranks"""
    from oracle.pyoracle import SEDOV
    for rank, ok, nbad, npeers, n_early in run(world, SEDOV, 16, 8, [0, 0, 0]):
        assert ok, f"rank {rank}: {nbad} ghost cells differ from the single-process fill"
        assert npeers >= 1


@pytest.mark.parametrize("world", [2, 4])
def test_overlapped_fill_completes_local_boxes_early(world):
    """Sedov octant, 32^3 in 8^3 boxes (64 boxes): boxes without remote ghost cells are complete (local copies + their
    physical boundaries) before the strips of the peers are waited for; the final state equals the blocking fill"""
    from oracle.pyoracle import SEDOV
    res = run(world, SEDOV, 32, 8, [0, 0, 0], overlapped=True)
    for rank, ok, nbad, npeers, n_early in res:
        assert ok, f"rank {rank}: {nbad} ghost cells differ from the single-process fill"
    assert sum(r[4] for r in res) > 0, "no rank had an early (remote-independent) box"


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_geometry_plans_are_consistent(world):
    """the exact decomposition bench.py uses at N GPUs (256^3 cells per GPU in 128^3 boxes, bricks of 8 boxes): every box owned once,
    8 boxes per rank, at most 7 peers, and for every pair of ranks what one side packs is what the other side expects (counts per
    peer and the ordered list of strips, which fixes the wire format)"""
    import bench
    from quokka_amd import capi
    from quokka_amd.multifab import Level, PlanningContext
    from quokka_amd.simulation import GhostExchange, Geometry, chop_domain, distribute_boxes
    n_cell = bench.weak_scaled_cells(256, world)
    geom = Geometry(3, n_cell, [0.0] * 3, [1.2 * n_cell[d] / 256 for d in range(3)], [0, 0, 0])
    boxes = chop_domain(n_cell, [128] * 3)
    owner = distribute_boxes(boxes, world, n_cell, [128] * 3)
    assert sorted(owner.count(r) for r in range(world)) == [8] * world
    bcs = [([capi.BC_REFLECT_EVEN] * 3, [capi.BC_REFLECT_EVEN] * 3)] * 6
    ctx = PlanningContext()
    plans = []
    for r in range(world):
        mine = [g for g, o in enumerate(owner) if o == r]
        lev = Level(ctx, 3, [boxes[g] for g in mine])
        ex = GhostExchange(lev, geom, 6, 4, boxes, owner, r, bcs)
        assert len(ex.peers) <= 7
        plans.append((mine, ex))
    for r, (mine, ex) in enumerate(plans):
        for k, peer, sbuf, rbuf in ex.peers:
            pmine, pex = plans[peer]
            kk = [q for q, (_, pr, _, _) in enumerate(pex.peers) if pr == r]
            assert len(kk) == 1, f"rank {peer} does not list rank {r} as a peer"
            _, _, psbuf, prbuf = pex.peers[kk[0]]
            assert sbuf.numel() == prbuf.numel() and rbuf.numel() == psbuf.numel()
            # strip by strip: (global dst box, global src box, region in the destination index space, shift, offset)
            sent = [(tuple(lo), tuple(hi), tuple(sh), off) for db, sb, lo, hi, sh, off in ex.items(1, k)]
            recv = [(tuple(lo), tuple(hi), tuple(sh), off) for db, sb, lo, hi, sh, off in pex.items(2, kk[0])]
            assert sent == recv, f"wire order differs between ranks {r} -> {peer}"


# ------------------------------------------------------------------------------------------------ ParallelCopy between two distributions
def pcopy_worker(rank, world, port, periodic, q):
    """A refined level distributed independently of its parent (quokka_amd/amr_simulation.py: CoarseShadow): the parent's valid cells reach
    the grown coarsened fine boxes (copy, periodic images included), and the one-cell ghost ring of the coarsened fine boxes is ADDED to the
    parent's valid cells (the fine part of a flux register on its way to the coarse owners).  numpy moves the bytes of the plan's items."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from quokka_amd.amr import ParallelCopy
        from quokka_amd.amr_simulation import chop_grids, distribute_sfc
        from quokka_amd.multifab import PlanningContext
        from quokka_amd.simulation import Geometry, chop_domain, distribute_boxes

        N, nc, ng = 32, 2, 3
        geom = Geometry(3, [N] * 3, [0.0] * 3, [1.0] * 3, list(periodic))
        pboxes = chop_domain(geom.n_cell, [16] * 3)
        powner = distribute_boxes(pboxes, world, geom.n_cell, [16] * 3)
        # shadow boxes (already coarsened): a block at the low corner and one that touches the high faces, chopped so that every rank owns some
        sboxes = chop_grids([([0, 0, 0], [15, 15, 7]), ([16, 24, 24], [31, 31, 31])], 2 * world, 16, 4, [N] * 3)
        sowner = distribute_sfc(sboxes, world, [powner.count(r) * 16 ** 3 for r in range(world)], unit=4)
        assert sorted(set(sowner)) == list(range(world))
        ctx = PlanningContext()

        def field(i, j, k, n):  # a function of the wrapped global index: what every copy of a cell must hold
            return ((i % N) + 100.0 * (j % N) + 10000.0 * (k % N)) * (n + 1)

        def make(boxes, owner, g, fill):
            out = []
            for (lo, hi), o in zip(boxes, owner):
                if o != rank:
                    continue
                beg = [lo[d] - g for d in range(3)]
                shp = [hi[d] - lo[d] + 1 + 2 * g for d in range(3)]
                k, j, i = np.meshgrid(*[np.arange(beg[d], beg[d] + shp[d]) for d in (2, 1, 0)], indexing="ij")
                a = np.stack([field(i, j, k, n) for n in range(nc)]) if fill else np.full((nc, shp[2], shp[1], shp[0]), np.nan)
                out.append((a, beg, (lo, hi)))
            return out

        def run(plan, src, dst, add):
            def pack(k, sbuf):
                buf = sbuf.numpy()
                for db, sb, lo, hi, sh, off in plan.items(1, k):
                    r = region(src[sb][0], src[sb][1], lo, hi, sh)
                    buf[off:off + r.size] = r.reshape(-1)

            def local():
                for db, sb, lo, hi, sh, off in plan.items(0):
                    r = region(dst[db][0], dst[db][1], lo, hi)
                    v = region(src[sb][0], src[sb][1], lo, hi, sh)
                    r[...] = r + v if add else v

            def unpack(k, rbuf):
                buf = rbuf.numpy()
                for db, sb, lo, hi, sh, off in plan.items(2, k):
                    r = region(dst[db][0], dst[db][1], lo, hi)
                    v = buf[off:off + r.size].reshape(r.shape)
                    r[...] = r + v if add else v
            plan.run(pack, local, unpack)

        # (1) parent valid -> shadow grown by 3
        parent = make(pboxes, powner, 0, True)
        shadow = make(sboxes, sowner, ng, False)
        fill = ParallelCopy(ctx, geom, pboxes, powner, sboxes, sowner, nc, rank, dst_nghost=ng)
        run(fill, parent, shadow, False)
        bad = 0
        for a, beg, (lo, hi) in shadow:
            k, j, i = np.meshgrid(*[np.arange(beg[d], beg[d] + a.shape[3 - d]) for d in (2, 1, 0)], indexing="ij")
            inside = np.ones(i.shape, dtype=bool)
            for d, x in enumerate((i, j, k)):
                if not periodic[d]:
                    inside &= (x >= 0) & (x < N)
            want = np.stack([field(i, j, k, n) for n in range(nc)])
            bad += int((a[:, inside] != want[:, inside]).sum()) + int((~np.isnan(a[:, ~inside])).sum())
        # (2) ring of the shadow boxes, added to the parent's valid cells: every parent cell ends up with (number of shadow boxes whose ring
        # holds the cell or a periodic image of it) x its own field value
        ring = make(sboxes, sowner, 1, True)
        for a, beg, (lo, hi) in ring:
            a[:, 1:-1, 1:-1, 1:-1] = 1.0e300  # valid cells of a shadow box must never travel
        acc = [(np.zeros_like(a), beg, b) for a, beg, b in make(pboxes, powner, 0, True)]
        add = ParallelCopy(ctx, geom, sboxes, sowner, pboxes, powner, nc, rank, src_nghost=1, src_ring_only=True)
        run(add, ring, acc, True)
        count = np.zeros((N + 2, N + 2, N + 2))
        for lo, hi in sboxes:
            g = np.zeros_like(count)
            g[lo[2]:hi[2] + 3, lo[1]:hi[1] + 3, lo[0]:hi[0] + 3] = 1.0
            g[lo[2] + 1:hi[2] + 2, lo[1] + 1:hi[1] + 2, lo[0] + 1:hi[0] + 2] = 0.0
            count += g
        c = count[1:-1, 1:-1, 1:-1].copy()  # index [k, j, i]; the rim of `count` is one cell beyond the domain: through a periodic face it wraps
        # (rims of corners through several periodic faces do not occur for these boxes with periodic = x only)
        if periodic[0]:
            c[:, :, N - 1] += count[1:-1, 1:-1, 0]
            c[:, :, 0] += count[1:-1, 1:-1, N + 1]
        for a, beg, (lo, hi) in acc:
            k, j, i = np.meshgrid(*[np.arange(lo[d], hi[d] + 1) for d in (2, 1, 0)], indexing="ij")
            want = np.stack([field(i, j, k, n) * c[k, j, i] for n in range(nc)])
            bad += int((a != want).sum())
        q.put((rank, bad, len(fill.peers), len(add.peers)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,periodic", [(2, [0, 0, 0]), (4, [1, 0, 0])])
def test_parallel_copy_and_add_between_independent_distributions(world, periodic):
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = free_port()
    procs = [mpctx.Process(target=pcopy_worker, args=(r, world, port, periodic, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, bad, npeers_fill, npeers_add in sorted(results):
        assert bad == 0, f"rank {rank}: {bad} values differ"
    assert sum(r[2] for r in results) > 0 and sum(r[3] for r in results) > 0, "nothing crossed ranks"
