"""The fused, x-vectorised CPU form of the hydro flux evaluation (oracle/hydro_fused.hpp: bench.py's cpu_baseline leg) against the operator-at-a-time
form of the same oracle (the line-by-line restatement of the reference that every GPU parity test is held to): equal in every bit — on the young
blast, on a developed Mach-3 shell that fires every limiter / flattening / HLLC-fan branch, on rough random states, and over whole runs."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle.pyoracle import K_B, M_U, SEDOV, Oracle  # noqa: E402


@pytest.fixture(scope="module")
def oracle():
    return Oracle("direct")


def mk(o, n, mgs):
    return o.sim(SEDOV, 3, [n] * 3, [0, 0, 0], [1.2] * 3, [0, 0, 0], max_grid_size=[mgs] * 3)


def same_bits(a, b):
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))


def check_fluxes(s, what):
    for b in range(s.nboxes):
        for d in range(3):
            F0, V0 = s.hydro_fluxes(b, d, fused=False)
            F1, V1 = s.hydro_fluxes(b, d, fused=True)
            assert not np.isnan(F0).any()
            assert same_bits(F0, F1) and same_bits(V0, V1), f"{what}: box {b} dir {d}: {int((F0 != F1).sum())} flux values differ"


def test_fused_fluxes_equal_the_operator_form_bit_for_bit(oracle):
    # (i) the young blast: 10 steps in, two box sizes (boxes that cut the hot region; 24^3: a row length that is no multiple of a vector)
    for n, mgs in ((32, 16), (24, 24)):
        s = mk(oracle, n, mgs)
        for _ in range(10):
            assert s.step()
        check_fluxes(s, f"blast {n}/{mgs}")
    # (ii) developed flow: the state tests/test_bench_geometry_gpu.py pins the GPU path to (a Mach-3 shell at 0.62 of the box edge, rippled)
    from quokka_amd.simulation import developed_state
    s = mk(oracle, 32, 16)
    for b in range(s.nboxes):
        lo, hi = s.box(b)
        s.set_state(developed_state(32, lo, hi), b)
    check_fluxes(s, "developed shell")
    # (iii) rough positive random states: every clamp, extremum and overshoot branch of PPM, all four fans of HLLC
    rng = np.random.default_rng(5)
    s = mk(oracle, 16, 16)
    shp = s.fab_shape(0)
    U = np.empty(shp)
    U[0] = rng.uniform(0.1, 10.0, shp[1:])
    for c in (1, 2, 3):
        U[c] = U[0] * rng.normal(0.0, 3.0, shp[1:])
    eint = rng.uniform(0.01, 50.0, shp[1:])
    U[4] = eint + 0.5 * (U[1] ** 2 + U[2] ** 2 + U[3] ** 2) / U[0]
    U[5] = eint * rng.uniform(0.5, 1.5, shp[1:])
    s.set_state(U, 0)
    check_fluxes(s, "random")


@pytest.mark.parametrize("stages", [False, True])
def test_whole_runs_in_the_fused_form_equal_the_operator_form(oracle, stages):
    """stages: the update, the limits and the dual-energy sync of a stage in the same pass over the box as the fluxes (HydroSim::fusedStage,
    fusedHydroUpdateBox), flux_rk2 accumulated while the fluxes are in registers; ghost cells compared too (the next step's stage inputs)"""
    a, b = mk(oracle, 32, 16), mk(oracle, 32, 16)
    b.set_fused_fluxes(True, stages=stages)
    for it in range(40):
        assert a.step() and b.step()
        assert a.dt == b.dt, it
    for k in range(a.nboxes):
        assert same_bits(a.state(k), b.state(k)), k
    assert a.counters() == b.counters()


@pytest.mark.parametrize("stages", [False, True])
def test_flux_correction_and_retries_from_the_fused_form(oracle, stages):
    """A 6x over-CFL step (tests/test_hydro_step_gpu.py::test_fofc_and_retries_match_oracle): PredictStep flags cells in both stages, the
    first-order flux correction runs and the advance is retried with dt / 2^n.  The fused stage hands such a stage to the operator path
    (which rewrites every cell the fused pass wrote): same flags, same counters, same state in every bit."""
    a, b = mk(oracle, 16, 8), mk(oracle, 16, 8)
    b.set_fused_fluxes(True, stages=stages)
    for _ in range(3):
        assert a.step() and b.step()
    dt = a.compute_dt() * 6.0
    assert b.compute_dt() * 6.0 == dt
    assert a.advance_fixed_dt(dt) and b.advance_fixed_dt(dt)
    co = a.counters()
    assert co["fofc1_cells"] > 0 and co["fofc2_cells"] > 0 and co["retries"] > 0, co
    assert b.counters() == co
    for _ in range(3):
        assert a.step() and b.step()
        assert a.dt == b.dt
    for k in range(a.nboxes):
        assert same_bits(a.state(k), b.state(k)), k


def test_limits_inside_the_fused_stage(oracle):
    """density and temperature floors that bite (EnforceLimits, reference src/hydro/hydro_system.hpp:696-773) inside the fused stage"""
    a, b = mk(oracle, 16, 16), mk(oracle, 16, 16)
    b.set_fused_fluxes(True, stages=True)
    for s in (a, b):
        s.set_limits(density_floor=0.9, temp_floor=1.0e-6 * M_U * 0.4 / K_B)  # (e = 1e-6: above the ambient gas, far below the blast)
    ref = mk(oracle, 16, 16)
    for _ in range(12):
        assert a.step() and b.step() and ref.step()
        assert a.dt == b.dt
    assert same_bits(a.state(0), b.state(0))
    U, R = a.state(0)[:, 4:-4, 4:-4, 4:-4], ref.state(0)[:, 4:-4, 4:-4, 4:-4]
    assert U[0].min() == 0.9 and R[0].min() < 0.9  # both floors did bite
    assert (U[5] / U[0]).min() >= 1.0e-6 * (1 - 1e-12) and (R[5] / R[0]).min() < 1.0e-7
