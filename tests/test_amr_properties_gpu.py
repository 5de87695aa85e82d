"""Properties of the AMR operators that AMReX owns in the reference (un-vendored: no AMReX here to compare with) — checked WITHOUT reference to the
oracle's restatement of the same formulas (tests/test_amr_ops_gpu.py compares GPU and oracle bit for bit; an error common to both would pass
there):
  * mf_linear_slope_minmax_interp (qk_InterpFromCoarse, method 1): exact for linear fields, conservative for every parent cell, bounded by the
    parent's 3x3x3 neighbourhood (no new extrema), constants reproduced exactly;
  * YAFluxRegister (qk_fluxreg_*): a brute-force conservation audit on a two-level hierarchy with RANDOM face fluxes — coarse and fine states are
    updated in numpy, only CrseAdd / FineAdd / Reflux and the average-down run on the GPU; the composite integral is conserved with the register
    and visibly not without it."""
import numpy as np
import pytest
import torch

from quokka_amd.amr import AverageDown, FluxRegister, InterpFromCoarse
from quokka_amd.multifab import Level, MultiFab
from quokka_amd.simulation import Geometry

pytestmark = pytest.mark.gpu


def _interp_setup(ctx, periodic=(1, 1, 1)):
    crse_boxes = [([0, 0, 0], [15, 15, 15]), ([16, 0, 0], [31, 15, 15])]
    fine_boxes = [([8, 8, 8], [23, 23, 23]), ([24, 8, 8], [47, 23, 23])]
    crse, fine = Level(ctx, 3, crse_boxes), Level(ctx, 3, fine_boxes)
    fgeom = Geometry(3, [64, 32, 32], [0.0] * 3, [2.0, 1.0, 1.0], list(periodic))
    return crse, fine, crse_boxes, fine_boxes, InterpFromCoarse(crse, fine, fgeom, 4, whole_fab=True)


def _coarse_fill(mf, fn):
    """fn(i, j, k) on the coarse index space (ghost cells included), periodic wrap left to the caller"""
    out = []
    for b in range(mf.level.nboxes):
        b0, shp = mf.begins[b], mf.shapes[b]
        k, j, i = np.meshgrid(*[np.arange(b0[d], b0[d] + shp[3 - d]) for d in (2, 1, 0)], indexing="ij")
        a = np.stack([fn(i, j, k, n) for n in range(shp[0])])
        mf.set_fab(b, a)
        out.append(a)
    return out


def test_interpolation_is_exact_for_linear_fields_and_constants(ctx):
    crse, fine, cb, fb, plan = _interp_setup(ctx, periodic=(0, 0, 0))
    nc = 3
    C0 = MultiFab(crse, nc, 4)
    coef = [(2.0, 0.0, 0.0, 0.0), (1.0, 0.25, -0.5, 0.125), (-3.0, 1.0, 1.0, 1.0)]  # constant; two linear fields
    _coarse_fill(C0, lambda i, j, k, n: coef[n][0] + coef[n][1] * (i + 0.5) + coef[n][2] * (j + 0.5) + coef[n][3] * (k + 0.5))
    F = MultiFab(fine, nc, 4, fill=float("nan"))
    plan(F, C0, C0, 1.0, 0.0, nc, 1, False)
    torch.cuda.synchronize()
    for b in range(fine.nboxes):
        got, b0 = F.fab_numpy(b), F.begins[b]
        k, j, i = np.meshgrid(*[np.arange(b0[d], b0[d] + got.shape[3 - d]) for d in (2, 1, 0)], indexing="ij")
        inside = (i >= 2) & (i < 62) & (j >= 2) & (j < 30) & (k >= 2) & (k < 30)  # (parents with a full stencil inside the coarse data)
        for n in range(nc):
            want = coef[n][0] + coef[n][1] * (i + 0.5) / 2 + coef[n][2] * (j + 0.5) / 2 + coef[n][3] * (k + 0.5) / 2  # fine centre in coarse index units
            assert np.isfinite(got[n][inside]).all()
            if n == 0:
                assert np.array_equal(got[n][inside], want[inside])  # a constant: exactly
            else:
                assert np.abs(got[n][inside] - want[inside]).max() <= 1e-13 * np.abs(want[inside]).max(), n


def test_interpolation_is_conservative_and_creates_no_new_extrema(ctx):
    crse, fine, cb, fb, plan = _interp_setup(ctx)
    rng = np.random.default_rng(21)
    nc = 2
    glob = [np.exp(rng.standard_normal((16, 16, 32))), rng.standard_normal((16, 16, 32)) * 10.0]  # a positive field with large jumps; a signed one
    C0 = MultiFab(crse, nc, 4)
    _coarse_fill(C0, lambda i, j, k, n: glob[n][k % 16, j % 16, i % 32])
    F = MultiFab(fine, nc, 4, fill=float("nan"))
    plan(F, C0, C0, 1.0, 0.0, nc, 1, False)
    torch.cuda.synchronize()
    nparents = 0
    for b in range(fine.nboxes):
        got, b0 = F.fab_numpy(b), F.begins[b]
        # fine fab = valid + 4 ghost cells: starts at an even index, even extent -> whole parents
        assert all(x % 2 == 0 for x in b0) and all(s % 2 == 0 for s in got.shape[1:])
        for n in range(nc):
            kids = got[n].reshape(got.shape[1] // 2, 2, got.shape[2] // 2, 2, got.shape[3] // 2, 2)
            pk, pj, pi = np.meshgrid(*[np.arange(b0[d] // 2, b0[d] // 2 + got.shape[3 - d] // 2) for d in (2, 1, 0)], indexing="ij")
            parent = glob[n][pk % 16, pj % 16, pi % 32]
            mean = kids.mean(axis=(1, 3, 5))
            assert np.abs(mean - parent).max() <= 4e-15 * np.abs(glob[n]).max(), n  # conservative: the children average to the parent
            lo, hi = parent.copy(), parent.copy()
            for dk in (-1, 0, 1):
                for dj in (-1, 0, 1):
                    for di in (-1, 0, 1):
                        v = glob[n][(pk + dk) % 16, (pj + dj) % 16, (pi + di) % 32]
                        lo, hi = np.minimum(lo, v), np.maximum(hi, v)
            tol = 4e-15 * np.abs(glob[n]).max()
            assert (kids.min(axis=(1, 3, 5)) >= lo - tol).all() and (kids.max(axis=(1, 3, 5)) <= hi + tol).all(), n  # bounded by the neighbourhood
            nparents += parent.size
    assert nparents > 5000


@pytest.mark.parametrize("with_reflux", [True, False])
def test_flux_register_conserves_the_composite_integral_with_random_fluxes(ctx, with_reflux):
    nc = 3
    dom_c, dom_f = [32, 16, 16], [64, 32, 32]
    crse_boxes = [([0, 0, 0], [15, 15, 15]), ([16, 0, 0], [31, 15, 15])]
    fine_boxes = [([0, 8, 8], [15, 23, 23]), ([16, 8, 8], [39, 23, 23]), ([24, 0, 0], [47, 7, 15])]
    crse, fine = Level(ctx, 3, crse_boxes), Level(ctx, 3, fine_boxes)
    cgeom = Geometry(3, dom_c, [0.0] * 3, [2.0, 1.0, 1.0], [1, 1, 1])
    dxc, dxf = cgeom.dx, [x / 2 for x in cgeom.dx]
    rng = np.random.default_rng(33)

    def global_faces(dom):
        """one random value per face of the periodic index space, per direction: (nc, nz, ny, nx) with face index = the cell to its right"""
        return [rng.standard_normal((nc, dom[2], dom[1], dom[0])) for _ in range(3)]

    def div(F, dx):
        """(1/dx)(F[i] - F[i+1]) summed over directions on the periodic index space"""
        out = np.zeros_like(F[0])
        for d in range(3):
            out += (F[d] - np.roll(F[d], -1, axis=3 - d)) / dx[d]
        return out

    def to_boxes(F, lev, boxes, dom):
        mfs = [MultiFab(lev, nc, 0, facedir=d) for d in range(3)]
        for d in range(3):
            for b, (lo, hi) in enumerate(boxes):
                idx = [np.arange(lo[e], hi[e] + 1 + (1 if e == d else 0)) % dom[e] for e in range(3)]
                mfs[d].set_fab(b, F[d][:, idx[2][:, None, None], idx[1][None, :, None], idx[0][None, None, :]])
        return mfs

    Fc, Ff1, Ff2 = global_faces(dom_c), global_faces(dom_f), global_faces(dom_f)
    Uc0 = rng.standard_normal((nc, dom_c[2], dom_c[1], dom_c[0])) + 5.0
    Uf0 = np.repeat(np.repeat(np.repeat(Uc0, 2, axis=1), 2, axis=2), 2, axis=3)  # fine data consistent with the coarse (piecewise constant)
    dtc = 0.013
    Uc = Uc0 + dtc * div(Fc, dxc)
    Uf = Uf0 + (dtc / 2) * div(Ff1, dxf) + (dtc / 2) * div(Ff2, dxf)
    covered = np.zeros(Uc0.shape[1:], dtype=bool)
    for lo, hi in fine_boxes:
        covered[lo[2] // 2:hi[2] // 2 + 1, lo[1] // 2:hi[1] // 2 + 1, lo[0] // 2:hi[0] // 2 + 1] = True
    fr = FluxRegister(crse, fine, cgeom, nc)
    fr.reset()
    fr.CrseAdd(to_boxes(Fc, crse, crse_boxes, dom_c), dxc, dtc)
    fr.FineAdd(to_boxes(Ff1, fine, fine_boxes, dom_f), dxf, dtc / 2)
    fr.FineAdd(to_boxes(Ff2, fine, fine_boxes, dom_f), dxf, dtc / 2)
    Uc_mf = MultiFab(crse, nc, 4, fill=0.0)
    for b, (lo, hi) in enumerate(crse_boxes):
        Uc_mf.valid(b).copy_(torch.from_numpy(np.ascontiguousarray(Uc[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1])))
    if with_reflux:
        fr.Reflux(Uc_mf)
    Uf_mf = MultiFab(fine, nc, 4, fill=0.0)
    for b, (lo, hi) in enumerate(fine_boxes):
        Uf_mf.valid(b).copy_(torch.from_numpy(np.ascontiguousarray(Uf[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1])))
    AverageDown(crse, fine)(Uf_mf, Uc_mf, 0, nc)
    torch.cuda.synchronize()
    after = np.zeros_like(Uc0)
    for b, (lo, hi) in enumerate(crse_boxes):
        after[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = Uc_mf.valid(b).cpu().numpy()
    # the composite integral in units of the coarse cell volume: level 0 after the average-down carries it
    for n in range(nc):
        drift = abs(after[n].sum() - Uc0[n].sum()) / abs(Uc0[n].sum())
        if with_reflux:
            assert drift <= 1e-14, (n, drift)
        else:
            assert drift > 1e-6, (n, drift)  # the coarse-fine flux mismatch of random fluxes is large: the audit can see a wrong register
    # cells far from the fine boxes are untouched by the register
    far = ~covered
    for lo, hi in fine_boxes:
        clo, chi = [x // 2 - 1 for x in lo], [x // 2 + 1 for x in hi]
        k, j, i = np.meshgrid(np.arange(16), np.arange(16), np.arange(32), indexing="ij")
        for sh in (-32, 0, 32):  # (periodic images in x; the boxes do not reach the y / z faces by more than one cell: check those too)
            near = (i + sh >= clo[0]) & (i + sh <= chi[0])
            near &= ((j >= clo[1]) & (j <= chi[1])) | ((j + 16 >= clo[1]) & (j + 16 <= chi[1])) | ((j - 16 >= clo[1]) & (j - 16 <= chi[1]))
            near &= ((k >= clo[2]) & (k <= chi[2])) | ((k + 16 >= clo[2]) & (k + 16 <= chi[2])) | ((k - 16 >= clo[2]) & (k - 16 <= chi[2]))
            far &= ~near
    assert far.sum() > 0 and np.array_equal(after[:, far], Uc[:, far])


def test_flux_register_conserves_the_composite_integral_in_two_dimensions(ctx):
    """the same brute-force audit on a 2-D hierarchy (AMREX_SPACEDIM = 2 builds: two fine faces per coarse face, four children per parent)"""
    nc, nd = 3, 2
    dom_c, dom_f = [32, 16, 1], [64, 32, 1]
    crse_boxes = [([0, 0, 0], [15, 15, 0]), ([16, 0, 0], [31, 15, 0])]
    fine_boxes = [([0, 8, 0], [15, 23, 0]), ([16, 8, 0], [39, 23, 0]), ([24, 0, 0], [47, 7, 0])]
    crse, fine = Level(ctx, 2, crse_boxes), Level(ctx, 2, fine_boxes)
    cgeom = Geometry(2, dom_c, [0.0] * 3, [2.0, 1.0, 1.0], [1, 1, 0])
    dxc = cgeom.dx
    dxf = [dxc[0] / 2, dxc[1] / 2, 1.0]
    rng = np.random.default_rng(34)

    def global_faces(dom):
        return [rng.standard_normal((nc, 1, dom[1], dom[0])) for _ in range(nd)]

    def div(F, dx):
        out = np.zeros_like(F[0])
        for d in range(nd):
            out += (F[d] - np.roll(F[d], -1, axis=3 - d)) / dx[d]
        return out

    def to_boxes(F, lev, boxes, dom):
        mfs = [MultiFab(lev, nc, 0, facedir=d) for d in range(nd)]
        for d in range(nd):
            for b, (lo, hi) in enumerate(boxes):
                idx = [np.arange(lo[e], hi[e] + 1 + (1 if e == d else 0)) % dom[e] for e in range(3)]
                mfs[d].set_fab(b, F[d][:, idx[2][:, None, None], idx[1][None, :, None], idx[0][None, None, :]])
        return mfs

    Fc, Ff1, Ff2 = global_faces(dom_c), global_faces(dom_f), global_faces(dom_f)
    Uc0 = rng.standard_normal((nc, 1, dom_c[1], dom_c[0])) + 5.0
    Uf0 = np.repeat(np.repeat(Uc0, 2, axis=2), 2, axis=3)
    dtc = 0.013
    Uc = Uc0 + dtc * div(Fc, dxc)
    Uf = Uf0 + (dtc / 2) * div(Ff1, dxf) + (dtc / 2) * div(Ff2, dxf)
    for with_reflux in (True, False):
        fr = FluxRegister(crse, fine, cgeom, nc, ratio=(2, 2, 1))
        fr.reset()
        fr.CrseAdd(to_boxes(Fc, crse, crse_boxes, dom_c), dxc, dtc)
        fr.FineAdd(to_boxes(Ff1, fine, fine_boxes, dom_f), dxf, dtc / 2)
        fr.FineAdd(to_boxes(Ff2, fine, fine_boxes, dom_f), dxf, dtc / 2)
        Uc_mf = MultiFab(crse, nc, 4, fill=0.0)
        for b, (lo, hi) in enumerate(crse_boxes):
            Uc_mf.valid(b).copy_(torch.from_numpy(np.ascontiguousarray(Uc[:, :, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1])))
        if with_reflux:
            fr.Reflux(Uc_mf)
        Uf_mf = MultiFab(fine, nc, 4, fill=0.0)
        for b, (lo, hi) in enumerate(fine_boxes):
            Uf_mf.valid(b).copy_(torch.from_numpy(np.ascontiguousarray(Uf[:, :, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1])))
        AverageDown(crse, fine, ratio=(2, 2, 1))(Uf_mf, Uc_mf, 0, nc)
        torch.cuda.synchronize()
        after = np.zeros_like(Uc0)
        for b, (lo, hi) in enumerate(crse_boxes):
            after[:, :, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = Uc_mf.valid(b).cpu().numpy()
        for n in range(nc):
            drift = abs(after[n].sum() - Uc0[n].sum()) / abs(Uc0[n].sum())
            assert (drift <= 1e-14) if with_reflux else (drift > 1e-6), (with_reflux, n, drift)


def test_flux_register_on_a_ring_of_fine_boxes_is_conservative_and_deterministic(ctx):
    """a refined RING (what a gradient criterion makes of a square pulse: Advection2D) has concave corners — coarse cells that touch fine faces of
    two directions, fine boxes that meet corner to corner: the brute-force audit of the 2-D test above on 12 fine boxes around a hole, twice"""
    nc, nd = 2, 2
    dom_c, dom_f = [64, 64, 1], [128, 128, 1]
    crse_boxes = [([16 * i, 16 * j, 0], [16 * i + 15, 16 * j + 15, 0]) for j in range(4) for i in range(4)]
    fine_boxes = [([16 * i, 16 * j, 0], [16 * i + 15, 16 * j + 15, 0]) for j in range(2, 6) for i in range(2, 6) if not (i in (3, 4) and j in (3, 4))]
    assert len(fine_boxes) == 12
    crse, fine = Level(ctx, 2, crse_boxes), Level(ctx, 2, fine_boxes)
    cgeom = Geometry(2, dom_c, [0.0] * 3, [1.0, 1.0, 1.0], [1, 1, 0])
    dxc = cgeom.dx
    dxf = [dxc[0] / 2, dxc[1] / 2, 1.0]
    rng = np.random.default_rng(35)
    Fc = [rng.standard_normal((nc, 1, dom_c[1], dom_c[0])) for _ in range(nd)]
    Ff = [[rng.standard_normal((nc, 1, dom_f[1], dom_f[0])) for _ in range(nd)] for _ in range(2)]

    def div(F, dx):
        out = np.zeros_like(F[0])
        for d in range(nd):
            out += (F[d] - np.roll(F[d], -1, axis=3 - d)) / dx[d]
        return out

    def to_boxes(F, lev, boxes, dom):
        mfs = [MultiFab(lev, nc, 0, facedir=d) for d in range(nd)]
        for d in range(nd):
            for b, (lo, hi) in enumerate(boxes):
                idx = [np.arange(lo[e], hi[e] + 1 + (1 if e == d else 0)) % dom[e] for e in range(3)]
                mfs[d].set_fab(b, F[d][:, idx[2][:, None, None], idx[1][None, :, None], idx[0][None, None, :]])
        return mfs

    Uc0 = rng.standard_normal((nc, 1, dom_c[1], dom_c[0])) + 5.0
    Uf0 = np.repeat(np.repeat(Uc0, 2, axis=2), 2, axis=3)
    dtc = 0.013
    Uc = Uc0 + dtc * div(Fc, dxc)
    Uf = Uf0 + (dtc / 2) * div(Ff[0], dxf) + (dtc / 2) * div(Ff[1], dxf)
    results = []
    for rep in range(2):
        fr = FluxRegister(crse, fine, cgeom, nc, ratio=(2, 2, 1))
        fr.reset()
        fr.CrseAdd(to_boxes(Fc, crse, crse_boxes, dom_c), dxc, dtc)
        for s in range(2):
            fr.FineAdd(to_boxes(Ff[s], fine, fine_boxes, dom_f), dxf, dtc / 2)
        Uc_mf = MultiFab(crse, nc, 4, fill=0.0)
        for b, (lo, hi) in enumerate(crse_boxes):
            Uc_mf.valid(b).copy_(torch.from_numpy(np.ascontiguousarray(Uc[:, :, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1])))
        fr.Reflux(Uc_mf)
        Uf_mf = MultiFab(fine, nc, 4, fill=0.0)
        for b, (lo, hi) in enumerate(fine_boxes):
            Uf_mf.valid(b).copy_(torch.from_numpy(np.ascontiguousarray(Uf[:, :, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1])))
        AverageDown(crse, fine, ratio=(2, 2, 1))(Uf_mf, Uc_mf, 0, nc)
        torch.cuda.synchronize()
        after = np.zeros_like(Uc0)
        for b, (lo, hi) in enumerate(crse_boxes):
            after[:, :, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = Uc_mf.valid(b).cpu().numpy()
        for n in range(nc):
            drift = abs(after[n].sum() - Uc0[n].sum()) / abs(Uc0[n].sum())
            assert drift <= 1e-14, (rep, n, drift)
        results.append(after)
    assert np.array_equal(results[0], results[1])
