"""The AMREX_SPACEDIM == 2 build of the hydro path (reference src/util/ArrayView_2d.hpp: the X2 view is an index SWAP, not the cyclic permutation
of the 3-D build; velocity components hydro_system.hpp:963-966).

CPU part (no GPU): the oracle's 2-D restatement is tied to its 3-D one, which the reference's known-answer tests pin: a problem uniform in z
(v_z = 0) performs the same arithmetic in both builds, so the x-y planes must agree bit for bit; HydroQuirk meets its own criterion in 2-D.
GPU part (-m gpu): the HIP operators in 2-D against the 2-D oracle, bit for bit."""
import numpy as np
import pytest

from oracle.pyoracle import BLAST2D, QUIRK


def quirk_delta_s(U):
    """test_quirk.cpp:128-182: |s(i0, j0 + 1) - s(i0, j0)|, s = P / rho^gamma, at (i0, j0) = (ishock_g, box lo) = (0, 0)"""
    g = 5.0 / 3.0
    P = lambda j: (g - 1.0) * (U[4, 0, j, 0] - 0.5 * (U[1, 0, j, 0] ** 2 + U[2, 0, j, 0] ** 2 + U[3, 0, j, 0] ** 2) / U[0, 0, j, 0])
    return abs(P(1) / U[0, 0, 1, 0] ** g - P(0) / U[0, 0, 0, 0] ** g)


def test_2d_build_equals_the_z_uniform_3d_build(oracle):
    N = 64
    s2 = oracle.sim(BLAST2D, 2, [N, N, 1], [0, 0, 0], [1.0, 1.0, 1.0], [0, 0, 0], max_grid_size=[32, 32, 1])
    s3 = oracle.sim(BLAST2D, 3, [N, N, 4], [0, 0, 0], [1.0, 1.0, 1.0], [0, 0, 0], max_grid_size=[32, 32, 4])
    assert s2.nboxes == s3.nboxes == 4
    for it in range(40):
        assert s2.step() and s3.step()
        assert s2.dt == s3.dt, it
    for b in range(4):
        a, c = s2.valid(b), s3.valid(b)
        assert np.array_equal(a[:, 0], c[:, 0]) and np.array_equal(c[:, 0], c[:, 3])
        assert np.all(c[3] == 0.0)
    U = np.zeros((6, N, N))
    for b in range(4):
        lo, hi = s2.box(b)
        U[:, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = s2.valid(b)[:, 0]
    # the blast has developed in both directions, with the symmetry of the square: x <-> y swaps the momenta
    assert np.abs(U[1]).max() > 1.0 and np.abs(U[2]).max() > 1.0
    assert np.allclose(U[0], U[0].T, rtol=1e-12) and np.allclose(U[1], U[2].T, rtol=1e-11, atol=1e-12)


def test_quirk_2d_meets_the_reference_criterion(oracle):
    """HydroQuirk as the 2-D build the reference's CMake enables it for (AMReX_SPACEDIM >= 2): max |delta s| <= 0.06 over the run
    (test_quirk.cpp:184-201), 772 steps of PLM + HLLC to t = 0.4; the sawtooth perturbation must decay (no carbuncle)"""
    s = oracle.sim(QUIRK, 2, [128, 16, 1], [0, 0, 0], [1.0, 0.125, 1.0], [0, 1, 1], max_grid_size=[128, 16, 1])
    dmax = 0.0
    while s.time < 0.4 and s.istep < 2000:
        assert s.step()
        dmax = max(dmax, quirk_delta_s(s.valid(0)))
    assert abs(s.time - 0.4) < 1e-14 and dmax <= 0.06
    U = s.valid(0)[:, 0]
    assert np.abs(U[2]).max() < 1e-10 * np.abs(U[1]).max()  # the sawtooth seeds no growing transverse flow (no carbuncle)
    assert np.array_equal(U[:, 0], U[:, 2]) and np.array_equal(U[:, 1], U[:, 3])  # the even rows stay identical, and so do the odd ones


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("order", [3, 2, 1])
def test_blast2d_steps_bit_exact_on_gpu(ctx, oracle, order, fused):
    """fused = True: the fused stage of a 2-D build — k_pre3 on the single plane (no z coefficient), x sweep, y sweep through the index-swap view
    carrying the epilogue — instead of the ~40 reference-shaped launches per stage"""
    from quokka_amd.simulation import blast2d_problem
    N, nsteps = 64, 25
    so = oracle.sim(BLAST2D, 2, [N, N, 1], [0, 0, 0], [1.0, 1.0, 1.0], [0, 0, 0], max_grid_size=[32, 32, 1], reconstruction_order=order)
    sg = blast2d_problem(ctx, N, 2, max_grid_size=[32, 32, 1], use_fused=fused)
    assert sg.use_fused == fused
    sg.reconstructionOrder_ = order
    assert so.nboxes == sg.lev.nboxes == 4
    for b in range(4):
        assert (so.box(b)[0], so.box(b)[1]) == (sg.my_boxes[b][0], sg.my_boxes[b][1])
        assert np.array_equal(so.valid(b), sg.state_new_cc_.valid(b).cpu().numpy())
    for it in range(nsteps):
        assert so.step() and sg.step(), it
        assert so.dt == sg.dt_, (it, so.dt, sg.dt_)
    for b in range(4):
        a, g = so.valid(b), sg.state_new_cc_.valid(b).cpu().numpy()
        assert np.abs(a[2]).max() > 0.5  # flow across the y faces
        assert np.array_equal(a, g), (b, np.abs(a - g).max())
    # every ghost cell of the reflecting walls and the box-box copies
    so.fill_ghosts(0, so.time)
    sg.fillBoundaryConditions(sg.state_new_cc_)
    for b in range(4):
        assert np.array_equal(so.state(b, 0), sg.state_new_cc_.fabs[b].cpu().numpy()), b


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_quirk_2d_on_gpu_matches_oracle_and_criterion(ctx, oracle, fused):
    from quokka_amd.simulation import quirk_problem
    so = oracle.sim(QUIRK, 2, [128, 16, 1], [0, 0, 0], [1.0, 0.125, 1.0], [0, 1, 1], max_grid_size=[128, 16, 1])
    sg = quirk_problem(ctx, 2, use_fused=fused)
    assert np.array_equal(so.valid(0), sg.state_new_cc_.valid(0).cpu().numpy())
    dmax = 0.0
    while sg.tNew_ < 0.4 and sg.istep < 2000:
        assert sg.step()
        if sg.istep <= 60:
            assert so.step() and so.dt == sg.dt_
            if sg.istep in (1, 10, 60):
                assert np.array_equal(so.valid(0), sg.state_new_cc_.valid(0).cpu().numpy()), sg.istep
        dmax = max(dmax, quirk_delta_s(sg.state_new_cc_.valid(0).cpu().numpy()))
    assert abs(sg.tNew_ - 0.4) < 1e-14 and dmax <= 0.06 and sg.istep == 772
