"""GPU parity of the data-parallel AMR pieces (C-ABI) against the CPU oracle: tags are integers (bit-exact by definition),
average-down sums in the same order (bit-exact)."""
import numpy as np
import pytest
import torch

from oracle.pyoracle import SEDOV
from quokka_amd import capi
from quokka_amd.amr import AverageDown, PostInterpState, PreInterpState, TagBoxArray, tag_centered_gradient, tag_relative_gradient
from quokka_amd.multifab import Level, MultiFab
from quokka_amd.simulation import sedov_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("field,eta,qmin,inclusive", [(capi.TAGFIELD_PRESSURE, 0.1, 1.0e-3, False), (0, 0.1, 1.0e-2, True), (4, 0.05, 0.0, False)])
def test_error_est_tags_match_oracle(ctx, oracle, field, eta, qmin, inclusive):
    """ErrorEst of HydroBlast3D (pressure, eta 0.1, P > 1e-3) and of RadhydroShell's form (density component, rho >= rho_min) on a Sedov
    state after 12 steps, 8 boxes: every tag"""
    N, mgs = 32, 16
    so = oracle.sim(SEDOV, 3, [N] * 3, [0, 0, 0], [1.2] * 3, [0, 0, 0], max_grid_size=[mgs] * 3)
    sg = sedov_problem(ctx, N, max_grid_size=mgs)
    for _ in range(12):
        assert so.step() and sg.step()
    so.fill_ghosts(0, so.time)
    sg.fillBoundaryConditions(sg.state_new_cc_)
    tags = TagBoxArray(sg.lev)
    tag_relative_gradient(sg.lev, sg.traits, sg.state_new_cc_, tags, field, eta, qmin, inclusive)
    nset = 0
    for b in range(so.nboxes):
        want = so.tag_relative_gradient(b, field, eta, qmin, inclusive)
        got = tags.fab_numpy(b)[0]
        assert got.dtype == np.int8 and np.array_equal(got, want), f"box {b}: {int((got != want).sum())} tags differ"
        nset += int((want == capi.TAG_SET).sum())
    assert 0 < nset < N ** 3, nset


@pytest.mark.parametrize("direction", [0, 1, 2])
def test_centered_gradient_tags_match_oracle(ctx, oracle, direction):
    """ErrorEst of HydroShocktube (centred density difference / (2 dx), rho >= 0.01) evaluated along each direction of a Sedov state"""
    N, mgs = 32, 16
    so = oracle.sim(SEDOV, 3, [N] * 3, [0, 0, 0], [1.2] * 3, [0, 0, 0], max_grid_size=[mgs] * 3)
    sg = sedov_problem(ctx, N, max_grid_size=mgs)
    for _ in range(12):
        assert so.step() and sg.step()
    so.fill_ghosts(0, so.time)
    sg.fillBoundaryConditions(sg.state_new_cc_)
    tags = TagBoxArray(sg.lev)
    dx, eta = 1.2 / N, 2.0
    tag_centered_gradient(sg.lev, sg.state_new_cc_, tags, 0, direction, dx, eta, 0.01, True)
    nset = 0
    for b in range(so.nboxes):
        want = so.tag_centered_gradient(b, 0, direction, dx, eta, 0.01, True)
        assert np.array_equal(tags.fab_numpy(b)[0], want)
        nset += int((want == capi.TAG_SET).sum())
    assert 0 < nset < N ** 3, nset


def test_average_down_matches_oracle(ctx, oracle):
    """two coarse boxes (16^3 each, side by side), three fine boxes straddling them (ratio 2): conservative mean, same summation order"""
    crse_boxes = [([0, 0, 0], [15, 15, 15]), ([16, 0, 0], [31, 15, 15])]
    fine_boxes = [([8, 8, 8], [39, 23, 23]), ([40, 8, 8], [55, 23, 23]), ([16, 24, 0], [39, 31, 15])]
    nc, ng = 6, 4
    crse, fine = Level(ctx, 3, crse_boxes), Level(ctx, 3, fine_boxes)
    rng = np.random.default_rng(3)
    U_c, U_f = MultiFab(crse, nc, ng), MultiFab(fine, nc, ng)
    c_np = [rng.standard_normal(s) for s in U_c.shapes]
    f_np = [rng.standard_normal(s) * 10.0 ** rng.integers(-3, 4) for s in U_f.shapes]
    for b, a in enumerate(c_np):
        U_c.set_fab(b, a)
    for b, a in enumerate(f_np):
        U_f.set_fab(b, a)
    avg = AverageDown(crse, fine)
    assert avg.num_items() == 5  # fine 0 overlaps both coarse boxes, fine 1 only the second, fine 2 both
    avg(U_f, U_c, 1, 4)
    torch.cuda.synchronize()
    want = [a.copy() for a in c_np]
    for fb, (flo, fhi) in enumerate(fine_boxes):
        for cb, (clo, chi) in enumerate(crse_boxes):
            lo = [max(flo[d] // 2, clo[d]) for d in range(3)]
            hi = [min(fhi[d] // 2, chi[d]) for d in range(3)]
            if all(lo[d] <= hi[d] for d in range(3)):
                oracle.average_down(f_np[fb], U_f.begins[fb], want[cb], U_c.begins[cb], (lo, hi), 1, 4)
    for b in range(2):
        got = U_c.fab_numpy(b)
        assert np.array_equal(got, want[b]), f"coarse box {b}: max diff {np.abs(got - want[b]).max()}"
        assert not np.array_equal(got, c_np[b])
    # conservative: the coarse sum over the region covered by fine box 1 is 1/8 of the fine sum (up to rounding)
    flo, fhi = fine_boxes[1]
    fsum = f_np[1][2, ng:-ng, ng:-ng, ng:-ng].sum()
    c1 = U_c.fab_numpy(1)[2]
    b0 = U_c.begins[1]
    csum = c1[flo[2] // 2 - b0[2]:fhi[2] // 2 - b0[2] + 1, flo[1] // 2 - b0[1]:fhi[1] // 2 - b0[1] + 1, flo[0] // 2 - b0[0]:fhi[0] // 2 - b0[0] + 1].sum()
    assert abs(csum - fsum / 8.0) <= 1e-12 * abs(fsum / 8.0) + 1e-12


def test_pre_post_interp_state(ctx):
    """PreInterpState: E -> (E - |p|^2 / (2 rho)) / rho on valid cells, PostInterpState: back; numpy restatement with the reference's
    association order (src/QuokkaSimulation.hpp:804-841), ghost cells untouched"""
    from test_hydro_ops_gpu import random_state
    boxes = [([0, 0, 0], [15, 11, 7]), ([16, 0, 0], [23, 11, 7])]
    lev = Level(ctx, 3, boxes)
    mf = MultiFab(lev, 6, 2)
    rng = np.random.default_rng(11)
    src = [random_state(rng, s[1:]) for s in mf.shapes]
    for b, a in enumerate(src):
        mf.set_fab(b, a)
    PreInterpState(lev, mf)
    pre = [mf.fab_numpy(b) for b in range(2)]
    PostInterpState(lev, mf)
    post = [mf.fab_numpy(b) for b in range(2)]
    for b, U in enumerate(src):
        v = (slice(None), slice(2, -2), slice(2, -2), slice(2, -2))
        rho, px, py, pz, E = (U[n] for n in range(5))
        ke = (px * px + py * py + pz * pz) / (2.0 * rho)
        want = U.copy()
        want[4][v[1:]] = ((E - ke) / rho)[v[1:]]
        assert np.array_equal(pre[b], want), f"PreInterpState, box {b}"
        back = want.copy()
        back[4][v[1:]] = (rho * want[4] + ke)[v[1:]]
        assert np.array_equal(post[b], back), f"PostInterpState, box {b}"
        assert np.allclose(post[b][4], U[4], rtol=1e-14)


@pytest.mark.parametrize("periodic", [[0, 0, 0], [1, 1, 0]])
@pytest.mark.parametrize("method,hooks,w", [(1, True, (0.25, 0.75)), (1, False, (1.0, 0.0)), (0, True, (0.5, 0.5))])
def test_interp_from_coarse_matches_oracle(ctx, oracle, periodic, method, hooks, w):
    """FillPatchTwoLevels' coarse part on a 2-level hierarchy: 32x16x16 coarse domain in two boxes, three fine boxes (one touching the
    low-x domain face).  (1) the plan covers every fine ghost cell that is inside the (periodic) domain and under no fine box exactly
    once; (2) the interpolated values equal the oracle's restatement bit for bit; (3) the mean of the 8 children equals the parent."""
    from quokka_amd.amr import InterpFromCoarse
    from quokka_amd.simulation import Geometry
    from test_hydro_ops_gpu import random_state
    ng, nc = 4, 6
    crse_boxes = [([0, 0, 0], [15, 15, 15]), ([16, 0, 0], [31, 15, 15])]
    fine_boxes = [([0, 8, 8], [15, 23, 23]), ([16, 8, 8], [39, 23, 23]), ([24, 0, 8], [47, 7, 15])]
    crse, fine = Level(ctx, 3, crse_boxes), Level(ctx, 3, fine_boxes)
    fgeom = Geometry(3, [64, 32, 32], [0.0] * 3, [2.0, 1.0, 1.0], periodic)
    plan = InterpFromCoarse(crse, fine, fgeom, ng)
    items = plan.items()
    # (1) coverage
    count = [np.zeros(s[1:], dtype=np.int32) for s in MultiFab(fine, 1, ng).shapes]
    Uf = MultiFab(fine, nc, ng, fill=-7.0)
    for fb, cb, lo, hi in items:
        b0 = Uf.begins[fb]
        count[fb][lo[2] - b0[2]:hi[2] - b0[2] + 1, lo[1] - b0[1]:hi[1] - b0[1] + 1, lo[0] - b0[0]:hi[0] - b0[0] + 1] += 1
    dom = [64, 32, 32]
    for fb, (flo, fhi) in enumerate(fine_boxes):
        b0 = Uf.begins[fb]
        kk, jj, ii = np.meshgrid(*[np.arange(b0[d], b0[d] + count[fb].shape[2 - d]) for d in (2, 1, 0)], indexing="ij")
        idx = [ii, jj, kk]
        inside = np.ones_like(ii, dtype=bool)
        for d in range(3):
            if not periodic[d]:
                inside &= (idx[d] >= 0) & (idx[d] < dom[d])
        covered = np.zeros_like(inside)
        for lo, hi in fine_boxes:
            for sx in (-1, 0, 1):
                for sy in (-1, 0, 1):
                    sh = [sx * dom[0] * periodic[0], sy * dom[1] * periodic[1], 0]
                    covered |= np.logical_and.reduce([(idx[d] >= lo[d] + sh[d]) & (idx[d] <= hi[d] + sh[d]) for d in range(3)])
        want = (inside & ~covered).astype(np.int32)
        assert np.array_equal(count[fb], want), f"fine box {fb}: plan covers {int(count[fb].sum())} cells, expected {int(want.sum())}"
    # (2) values
    rng = np.random.default_rng(5)
    Co, Cn = MultiFab(crse, nc, ng), MultiFab(crse, nc, ng)
    co_np = [random_state(rng, s[1:]) for s in Co.shapes]
    cn_np = [a * (1.0 + 0.05 * rng.standard_normal(a.shape)) for a in co_np]
    for b in range(2):
        Co.set_fab(b, co_np[b])
        Cn.set_fab(b, cn_np[b])
    plan(Uf, Co, Cn, w[0], w[1], nc, method, hooks)
    torch.cuda.synchronize()
    want = [np.full(s, -7.0) for s in Uf.shapes]
    for fb, cb, lo, hi in items:
        oracle.interp_from_coarse(want[fb], Uf.begins[fb], co_np[cb], cn_np[cb], Co.begins[cb], (lo, hi), w[0], w[1], nc, method, hooks)
    for fb in range(3):
        got = Uf.fab_numpy(fb)
        assert np.array_equal(got, want[fb]), f"fine box {fb}: max diff {np.nanmax(np.abs(got - want[fb]))}"
    # (3) conservative: density children average to the (time-interpolated) parent
    fb, cb, lo, hi = max(items, key=lambda it: np.prod([it[3][d] - it[2][d] + 1 for d in range(3)]))
    if all((hi[d] - lo[d] + 1) >= 2 for d in range(3)):
        l2 = [lo[d] + (lo[d] % 2) for d in range(3)]
        b0, c0 = Uf.begins[fb], Co.begins[cb]
        kids = Uf.fab_numpy(fb)[0, l2[2] - b0[2]:l2[2] - b0[2] + 2, l2[1] - b0[1]:l2[1] - b0[1] + 2, l2[0] - b0[0]:l2[0] - b0[0] + 2]
        par_o = co_np[cb][0, l2[2] // 2 - c0[2], l2[1] // 2 - c0[1], l2[0] // 2 - c0[0]]
        par_n = cn_np[cb][0, l2[2] // 2 - c0[2], l2[1] // 2 - c0[1], l2[0] // 2 - c0[0]]
        parent = par_o if w[1] == 0.0 else w[0] * par_o + w[1] * par_n
        assert abs(kids.mean() - parent) <= 1e-13 * abs(parent)


@pytest.mark.parametrize("periodic", [[0, 0, 0], [1, 0, 1]])
def test_flux_register_matches_numpy_restatement(ctx, periodic):
    """YAFluxRegister: one CrseAdd (dt_c), two FineAdds (dt_f = dt_c / 2), Reflux onto a zero coarse state, against a numpy
    restatement that walks the same items (the item set itself is checked independently: every coarse cell that is face-adjacent to a
    fine box, not under a fine box and inside the (periodic) domain appears once per adjacent face)."""
    from quokka_amd.amr import FluxRegister
    from quokka_amd.simulation import Geometry
    nc = 6
    crse_boxes = [([0, 0, 0], [15, 15, 15]), ([16, 0, 0], [31, 15, 15])]
    fine_boxes = [([0, 8, 8], [15, 23, 23]), ([16, 8, 8], [39, 23, 23]), ([24, 0, 0], [47, 7, 15])]
    crse, fine = Level(ctx, 3, crse_boxes), Level(ctx, 3, fine_boxes)
    cgeom = Geometry(3, [32, 16, 16], [0.0] * 3, [2.0, 1.0, 1.0], periodic)
    dxc, dxf = cgeom.dx, [x / 2 for x in cgeom.dx]
    fr = FluxRegister(crse, fine, cgeom, nc)
    items = fr.items()
    # --- the item set
    dom = [32, 16, 16]
    cf = [([lo[d] // 2 for d in range(3)], [hi[d] // 2 for d in range(3)]) for lo, hi in fine_boxes]
    under_fine = np.zeros((16 + 4, 16 + 4, 32 + 4), dtype=bool)  # index + 2 (wrapped images included)

    def wrap(i, d):
        return i % dom[d] if periodic[d] else i

    expect = set()
    for fb, (lo, hi) in enumerate(cf):
        for d in range(3):
            for side in (0, 1):
                plane = lo[d] - 1 if side == 0 else hi[d] + 1
                rng = [range(lo[e], hi[e] + 1) for e in range(3)]
                rng[d] = [plane]
                for k in rng[2]:
                    for j in rng[1]:
                        for i in rng[0]:
                            c = [i, j, k]
                            w = [wrap(c[e], e) for e in range(3)]
                            if any(w[e] < 0 or w[e] >= dom[e] for e in range(3)):
                                continue
                            if any(all(l2[e] <= w[e] <= h2[e] for e in range(3)) for l2, h2 in cf):
                                continue
                            expect.add((d, side, fb, tuple(c)))
    got = set()
    for d, side, fb, cb, lo, hi, sh in items:
        for k in range(lo[2], hi[2] + 1):
            for j in range(lo[1], hi[1] + 1):
                for i in range(lo[0], hi[0] + 1):
                    key = (d, side, fb, (i, j, k))
                    assert key not in got
                    got.add(key)
                    w = (i + sh[0], j + sh[1], k + sh[2])
                    clo, chi = crse_boxes[cb]
                    assert all(clo[e] <= w[e] <= chi[e] for e in range(3))
    assert got == expect, (len(got), len(expect))
    # --- the arithmetic
    rng_ = np.random.default_rng(9)
    Fc = [MultiFab(crse, nc, 0, facedir=d) for d in range(3)]
    Ff1 = [MultiFab(fine, nc, 0, facedir=d) for d in range(3)]
    Ff2 = [MultiFab(fine, nc, 0, facedir=d) for d in range(3)]
    host = {}
    for name, mfs in (("c", Fc), ("f1", Ff1), ("f2", Ff2)):
        for d in range(3):
            for b in range(mfs[d].level.nboxes):
                a = rng_.standard_normal(mfs[d].shapes[b])
                mfs[d].set_fab(b, a)
                host[(name, d, b)] = a
    U = MultiFab(crse, nc, 4, fill=0.0)
    dtc = 0.37
    fr.reset()
    fr.CrseAdd(Fc, dxc, dtc)
    fr.FineAdd(Ff1, dxf, dtc / 2)
    fr.FineAdd(Ff2, dxf, dtc / 2)
    fr.Reflux(U)
    torch.cuda.synchronize()
    want = [np.zeros(s) for s in U.shapes]
    for g in range(6):  # Reflux applies the (dir, side) groups in order
        for d, side, fb, cb, lo, hi, sh in items:
            if 2 * d + side != g:
                continue
            cb0, fb0 = Fc[d].begins[cb], Ff1[d].begins[fb]
            a1, a2 = sorted([(d + 1) % 3, (d + 2) % 3])
            for k in range(lo[2], hi[2] + 1):
                for j in range(lo[1], hi[1] + 1):
                    for i in range(lo[0], hi[0] + 1):
                        o = [i, j, k]
                        f = [o[e] + sh[e] for e in range(3)]
                        if side == 0:
                            f[d] += 1
                        reg = np.zeros(nc)
                        v = (dtc / dxc[d]) * host[("c", d, cb)][:, f[2] - cb0[2], f[1] - cb0[1], f[0] - cb0[0]]
                        reg = reg + v if side == 0 else reg - v
                        for name in ("f1", "f2"):
                            base = [o[e] * 2 for e in range(3)]
                            base[d] = (o[d] + 1) * 2 if side == 0 else o[d] * 2
                            ssum = np.zeros(nc)
                            for q in range(2):
                                for p in range(2):
                                    ff = list(base)
                                    ff[a1] += p
                                    ff[a2] += q
                                    ssum = ssum + host[(name, d, fb)][:, ff[2] - fb0[2], ff[1] - fb0[1], ff[0] - fb0[0]]
                            v = ((dtc / 2) / (dxf[d] * 8.0)) * ssum
                            reg = reg - v if side == 0 else reg + v
                        u0 = U.begins[cb]
                        w = [o[e] + sh[e] for e in range(3)]
                        want[cb][:, w[2] - u0[2], w[1] - u0[1], w[0] - u0[0]] += reg
    for b in range(2):
        got_u = U.fab_numpy(b)
        assert np.array_equal(got_u, want[b]), f"coarse box {b}: max diff {np.abs(got_u - want[b]).max()}"
    assert sum(int((w != 0).sum()) for w in want) > 0


def test_flux_register_save_restore_for_retries(ctx):
    """advanceHydroAtLevelWithRetries (reference src/QuokkaSimulation.hpp:894-929): a failed attempt of the FINE level whose first substep
    succeeded has already been added to the register; restore() must bring back exactly the state save() saw (coarse contribution + the
    fine substeps of earlier, successful advances), so that Reflux counts every flux once."""
    from quokka_amd.amr import FluxRegister
    from quokka_amd.simulation import Geometry
    nc = 6
    crse_boxes = [([0, 0, 0], [15, 15, 15])]
    fine_boxes = [([8, 8, 8], [23, 23, 23])]
    crse, fine = Level(ctx, 3, crse_boxes), Level(ctx, 3, fine_boxes)
    cgeom = Geometry(3, [16, 16, 16], [0.0] * 3, [1.0, 1.0, 1.0], [0, 0, 0])
    dxc, dxf = cgeom.dx, [x / 2 for x in cgeom.dx]
    rng_ = np.random.default_rng(3)

    def rand_faces(lev):
        mfs = [MultiFab(lev, nc, 0, facedir=d) for d in range(3)]
        for d in range(3):
            for b in range(lev.nboxes):
                mfs[d].set_fab(b, rng_.standard_normal(mfs[d].shapes[b]))
        return mfs

    Fc, Ff1, Ff2, Fbad = rand_faces(crse), rand_faces(fine), rand_faces(fine), rand_faces(fine)

    def reflux_of(sequence):
        fr = FluxRegister(crse, fine, cgeom, nc)
        fr.reset()
        sequence(fr)
        U = MultiFab(crse, nc, 4, fill=0.0)
        fr.Reflux(U)
        return U.fab_numpy(0)

    def clean(fr):  # coarse step, first fine step, second fine step: no retry anywhere
        fr.CrseAdd(Fc, dxc, 0.4)
        fr.FineAdd(Ff1, dxf, 0.2)
        fr.FineAdd(Ff2, dxf, 0.2)

    def with_retry(fr):  # the second fine step fails after its first half-substep was added, is retried and succeeds
        fr.CrseAdd(Fc, dxc, 0.4)
        fr.FineAdd(Ff1, dxf, 0.2)
        fr.save()
        fr.FineAdd(Fbad, dxf, 0.1)  # substep 1 of the failed attempt (substep 2 failed: nothing added)
        fr.restore()
        fr.FineAdd(Ff2, dxf, 0.2)

    want, got = reflux_of(clean), reflux_of(with_retry)
    assert np.abs(want).max() > 0.0
    assert np.array_equal(want, got)
