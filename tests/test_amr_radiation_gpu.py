"""Radiation on refined levels (quokka_amd/amr_simulation.py::RadAmrLevelSim; reference src/QuokkaSimulation.hpp:653-707, :1577-1860) through
size-independent properties: a refined level that covers the whole domain reproduces the uniform fine-grid radiation run bit for bit; with
partial refinement the composite radiation + gas energy is conserved to rounding only if the radiation flux registers do their job."""
import numpy as np
import pytest

from quokka_amd import capi
from quokka_amd.amr_simulation import rad_pulse_amr_problem
from quokka_amd.radhydro import RadhydroSimulation
from quokka_amd.simulation import Geometry, chop_domain

pytestmark = pytest.mark.gpu


def uniform_twin(ctx, amr, n, mgs):
    """the same problem on a uniform grid of n^3 cells (the level-1 resolution), initialised by the same function"""
    import torch
    geom = Geometry(3, [n, n, n], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [1, 1, 1])
    sim = RadhydroSimulation(ctx, geom, amr.traits, amr.rad_traits, amr.bcs, [mgs] * 3)
    sim.is_hydro_enabled = amr.is_hydro_enabled
    sim.radiationCflNumber_, sim.maxSubsteps_, sim.radiationReconstructionOrder_ = amr.radiationCflNumber_, amr.maxSubsteps_, amr.radiationReconstructionOrder_
    sim.set_initial_conditions(amr.initial_conditions(geom))
    return sim


@pytest.mark.parametrize("hydro", [False, True])
def test_full_coverage_equals_the_uniform_fine_radiation_run(ctx, hydro):
    N = 16
    fine_boxes = chop_domain([2 * N] * 3, [16] * 3)
    amr = rad_pulse_amr_problem(ctx, N, 1, max_grid_size=16, static_fine_boxes=[fine_boxes], hydro=hydro)
    assert amr.finest_level == 1 and amr.levels[1].fluxreg_rad.items() == []
    uni = uniform_twin(ctx, amr, 2 * N, 16)
    f = amr.levels[1]
    for b in range(f.lev.nboxes):
        assert np.array_equal(f.state_new_cc_.valid(b).cpu().numpy(), uni.state_new_cc_.valid(b).cpu().numpy())
    for it in range(4):
        amr.step()
        for _ in range(2):
            assert uni.step(amr.dt_[1])
        for b in range(f.lev.nboxes):
            assert np.array_equal(f.state_new_cc_.valid(b).cpu().numpy(), uni.state_new_cc_.valid(b).cpu().numpy()), f"step {it}, box {b}"
    # the pulse has moved energy into the gas and spread: the test looks at a developed state
    U = f.state_new_cc_.valid(0).cpu().numpy()
    assert np.abs(U[7:10]).max() > 1e-3 and U[5].max() > 1.5 + 1e-4
    # level 0 = the average of level 1, radiation block included
    c = amr.levels[0]
    fine = np.zeros((10, 2 * N, 2 * N, 2 * N))
    for (lo, hi), v in zip(f.my_boxes, f.gather_valid_local()):
        fine[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = v
    crse = np.zeros((10, N, N, N))
    for (lo, hi), v in zip(c.my_boxes, c.gather_valid_local()):
        crse[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = v
    avg = fine.reshape(10, N, 2, N, 2, N, 2).mean(axis=(2, 4, 6))
    for n in (0, 4, 6):
        assert np.allclose(crse[n], avg[n], rtol=1e-13, atol=1e-300), n


@pytest.mark.parametrize("reflux", [True, False])
def test_partial_refinement_conserves_radiation_plus_gas_energy_with_reflux(ctx, reflux):
    """static 2-level hierarchy: the centre of a 32^3 periodic box refined (32^3 fine cells over the central 16^3 coarse cells).  The pulse
    crosses the coarse-fine interface; E_int + (c / c_hat) E_rad of the composite grid stays constant to the tolerance of the Newton-Raphson
    solves with the radiation flux register, and drifts by orders of magnitude more without Reflux.  (The INTERNAL energy: with beta_order 0
    the gas collects the momentum the radiation deposits without a work term, so the kinetic part of the total energy is not balanced — on
    a uniform grid either.)"""
    N = 32
    amr = rad_pulse_amr_problem(ctx, N, 1, max_grid_size=32, static_fine_boxes=[[([16, 16, 16], [47, 47, 47])]])
    amr.do_reflux = reflux
    assert len(amr.levels[1].cf_interp.items()) > 0 and len(amr.levels[1].fluxreg_rad.items()) == 6
    total = lambda: amr.composite_sum(5) + amr.composite_sum(6)
    e0, r0 = total(), amr.composite_sum(6)
    for _ in range(40):
        amr.step()
    e1, r1 = total(), amr.composite_sum(6)
    # radiation has left the refined region (the coarse cells far from the centre are above the background) and heated the gas
    c = amr.levels[0]
    far = c.state_new_cc_.valid(0)[6][2, 16, 16].item()
    assert far > 1.0 + 1e-4 and r1 < r0 - 1e-4
    drift = abs(e1 - e0) / e0
    if reflux:
        assert drift < 5e-11, drift  # (40 steps x 2 solves at a residual tolerance of 1e-11 of the cell's energy, random sign)
    else:
        assert drift > 1e-8, drift


def test_refined_pulse_follows_the_uniform_fine_solution(ctx):
    """the composite solution against the uniform run at the fine resolution: inside the refined region the radiation energy agrees to a few
    per mille (the difference is the coarse data interpolated into the ghost cells of the level), the coarse level's far field to a few per cent
    of the excess"""
    N = 32
    amr = rad_pulse_amr_problem(ctx, N, 1, max_grid_size=32, static_fine_boxes=[[([16, 16, 16], [47, 47, 47])]])
    uni = uniform_twin(ctx, amr, 2 * N, 64)
    for _ in range(20):
        amr.step()
        for _ in range(2):
            assert uni.step(amr.dt_[1])
    f = amr.levels[1].state_new_cc_.valid(0).cpu().numpy()
    u = uni.state_new_cc_.valid(0).cpu().numpy()[:, 16:48, 16:48, 16:48]
    assert np.abs(f[6] - u[6]).sum() / np.abs(u[6] - 1.0).sum() < 0.02
    assert np.abs(f[5] - u[5]).sum() / np.abs(u[5] - 1.5).sum() < 0.02


def test_dynamic_regridding_follows_the_pulse_and_conserves(ctx):
    """tags from the radiation energy, Berger-Rigoutsos grids, regrid every 2 coarse steps: new fine cells are interpolated conservatively from
    the coarse level with the radiation block, so E_int + E_rad of the composite grid stays at the Newton tolerance while the grids change"""
    amr = rad_pulse_amr_problem(ctx, 32, 1, max_grid_size=32, blocking_factor=8)
    assert amr.finest_level == 1
    # (a low threshold: the pulse stays refined while it is steep.  De-refining a pulse the coarse grid cannot resolve makes the transport
    # step repair radiation states — amendRadState, E below its floor or |F| > c E after a PPM overshoot — which does not conserve, AMR or not)
    amr.tag_threshold = 1.02
    total = lambda: amr.composite_sum(5) + amr.composite_sum(6)
    e0 = total()
    grids = {str(sorted(map(str, amr.levels[1].all_boxes)))}
    cells = [amr.CountCells(1)]
    for _ in range(30):
        amr.step()
        if amr.finest_level >= 1:
            grids.add(str(sorted(map(str, amr.levels[1].all_boxes))))
            cells.append(amr.CountCells(1))
    assert len(grids) > 1, "the refined region never changed"
    assert abs(total() - e0) / e0 < 1e-10
    # tagged cells are inside the refined region at the end (n_error_buf = 3 cells of margin)
    L0, tagged_uncovered = amr.levels[0], 0
    if amr.finest_level >= 1:
        cover = np.zeros((32, 32, 32), dtype=bool)
        for lo, hi in amr.levels[1].all_boxes:
            cover[lo[2] // 2:hi[2] // 2 + 1, lo[1] // 2:hi[1] // 2 + 1, lo[0] // 2:hi[0] // 2 + 1] = True
        E = L0.state_new_cc_.valid(0)[6].cpu().numpy()
        tagged_uncovered = int(((E > amr.tag_threshold) & ~cover).sum())
    assert tagged_uncovered == 0


@pytest.mark.parametrize("hydro", [False, True])
def test_fused_radiation_stage_on_the_boxes_of_a_hierarchy(ctx, hydro):
    """qk_rad_stage_fused against the separate operators on boxes that are not cubes: a static two-level hierarchy around the pulse whose refined level is an
    L of three boxes 40 x 8 x 16, 8 x 24 x 16 and 16 x 16 x 8 fine cells (the shapes a blocking factor of 8 produces), level 0 in 16^3 boxes.  Both levels with
    the fused stage, then both with computeRadiationFluxes + PredictStep / AddFluxesRK2: four coarse steps (subcycled radiation, flux registers of the
    radiation block fed by either form), every component of every level equal in every bit."""
    fine = [([24, 24, 24], [63, 31, 39]), ([24, 32, 24], [31, 55, 39]), ([32, 32, 32], [47, 47, 39])]
    runs = []
    for fused in (True, False):
        amr = rad_pulse_amr_problem(ctx, 32, 1, max_grid_size=16, static_fine_boxes=[fine], hydro=hydro)
        assert amr.finest_level == 1 and [tuple(map(tuple, b)) for b in amr.levels[1].all_boxes] == [tuple(map(tuple, b)) for b in fine]
        for L in amr.levels:
            assert L.use_fused_rad
            L.use_fused_rad = fused
        for _ in range(4):
            amr.step()
        runs.append([[v.copy() for v in L.gather_valid_local()] for L in amr.levels])
    for la, lb in zip(*runs):
        for a, b in zip(la, lb):
            assert np.isfinite(a).all() and np.array_equal(a, b)
    U = runs[0][1][0]
    assert np.abs(U[7:10]).max() > 1e-4  # (the pulse has reached the refined boxes: fluxes are not zero there)
