"""Pins the MULTIGROUP radiation restatement of the CPU oracle (oracle/radiation_multigroup.hpp, oracle/problems_multigroup.hpp):
the helper functions against independent formulas (quadrature of the definitions), the Planck table against the reference's own listing
(in this container only: /root/reference is read in place, nothing is copied), and the coupled scheme against the pass criteria of the
reference's multigroup ctests."""
import os
import re

import numpy as np
import pytest

from oracle.pyoracle import (LINE_COOLING, LINE_COOLING_MG, MARSHAK_DUST, MARSHAK_DUST_PE, MARSHAK_VAYTET, PIECEWISE_CONSTANT, PPL_FIXED_SLOPE, PPL_FULL_SPECTRUM, PULSE_MG, PULSE_MG_GREY, RADDUST, RADDUST_MG,
                             RADSHOCK_MG, RADTUBE)

HERE = os.path.dirname(os.path.abspath(__file__))
K_B, H_PLANCK, C_LIGHT = 1.380649e-16, 6.62607015e-27, 2.99792458e10
A_RAD = 4.0 * 5.670374419e-5 / C_LIGHT
REF_PLANCK = "/root/reference/src/radiation/planck_integral.hpp"


def planck_Y(x):
    """(15/pi^4) int_0^x t^3/(e^t - 1) dt by the series  sum_n e^{-nx} (x^3/n + 3x^2/n^2 + 6x/n^3 + 6/n^4)  for the complement"""
    import mpmath as mp
    mp.mp.dps = 30
    return float(15 / mp.pi ** 4 * mp.quad(lambda t: t ** 3 / mp.expm1(t), [0, min(x, 1.0), x] if x > 1 else [0, x]))


def test_planck_table_is_the_function_it_says(oracle):
    tab = oracle.planck_table()
    assert tab.shape == (1000,) and np.all(np.diff(tab) >= 0) and tab[-1] == 1.0
    for j in (0, 1, 137, 500, 640, 777, 900, 999):
        x = 10.0 ** (-3 + j * 5 / 999)
        assert abs(tab[j] - planck_Y(x)) <= 1e-14 * tab[j], j  # (x itself is rounded here: Y ~ x^3 carries 3 ulp of it)
    # two computations of one function: the oracle's own table (oracle/planck_table.hpp: Bernoulli series / exponential series in 113-bit
    # arithmetic, made when liboracle.so loads) and the data file the product is built from (tools/make_planck_table.py: mpmath quadrature at
    # 50 digits) — every one of the 1000 doubles is the correctly rounded value of the function, so they are equal in every bit
    txt = open(os.path.join(os.path.dirname(HERE), "quokka_amd", "data", "planck_integral_table.inc")).read()
    vals = np.array([float(v) for v in re.findall(r"^([0-9.e+-]+),$", txt, flags=re.M)])
    assert np.array_equal(vals, tab)


@pytest.mark.skipif(not os.path.exists(REF_PLANCK), reason="the reference tree is only present in the build container")
def test_planck_table_agrees_with_the_reference_listing(oracle):
    """The reference lists the same table to 15 significant digits; ours is computed from the definition to 17.  They must describe the
    same function: the largest relative difference is 5e-14 (the reference's last printed digit is not correctly rounded everywhere)."""
    src = open(REF_PLANCK).read()
    body = src[src.index("Y_interp = {") + len("Y_interp = {"):]
    body = body[:body.index("};")]
    ref = np.array([float(v) for v in re.findall(r"[0-9]\.[0-9]+e[+-][0-9]+", body)])
    tab = oracle.planck_table()
    assert ref.shape == tab.shape == (1000,)
    assert np.max(np.abs(ref - tab) / tab) < 1e-13


def test_planck_integral_branches(oracle):
    # below the table: second-order series, clamped to the first table entry; above: exactly 1; inside: linear interpolation in log10 x
    assert oracle.planck_integral(0.0) == 0.0 and oracle.planck_integral(-1.0) == 0.0
    assert oracle.planck_integral(100.0) == 1.0 and oracle.planck_integral(1e3) == 1.0
    x = 5e-4
    assert oracle.planck_integral(x) == min((-4 + x) * x + 8 * np.log((2 + x) / 2), oracle.planck_table()[0])
    for x in (2e-3, 0.37, 1.0, 2.8214, 9.9, 57.0):
        assert abs(oracle.planck_integral(x) - planck_Y(x)) < 1e-4 * planck_Y(x) + 1e-12  # the interpolation error of the reference's scheme
    xs = np.logspace(-2.99, 1.99, 4001)
    ys = np.array([oracle.planck_integral(v) for v in xs])
    assert np.all(np.diff(ys) >= 0)


def test_planck_fractions_sum_to_one_and_floor(oracle):
    b = np.array([1e15, 1e16, 1e17, 1e18, 1e19, 1e20])
    T = 2.18e6
    f, E = oracle.planck_fractions(b, H_PLANCK, K_B, A_RAD, 0.0, T)
    assert abs(f.sum() - 1.0) < 1e-15 and np.all(f >= 0)
    # the first group takes everything below its upper edge, the last everything above its lower edge (radiation_system.hpp:430-461)
    x = b[1:-1] * H_PLANCK / (K_B * T)
    cum = np.array([planck_Y(v) for v in x])
    assert np.allclose(np.cumsum(f)[:-1], cum, rtol=1e-4, atol=1e-14)
    assert np.allclose(E, A_RAD * T ** 4 * f, rtol=1e-15)
    floor = 1e-3 * A_RAD * T ** 4
    _, Ef = oracle.planck_fractions(b, H_PLANCK, K_B, A_RAD, floor, T)
    assert np.all(Ef >= floor / 5) and Ef[4] == floor / 5  # Erad_floor_ = Erad_floor / nGroups (:211)


def test_planck_function_is_the_derivative_of_the_integral(oracle):
    T = 1.0e7
    for nu in (1e15, 3e17, 1.2e18, 9e18):
        x = H_PLANCK * nu / (K_B * T)
        expect = (H_PLANCK / (K_B * T)) * (15 / np.pi ** 4) * A_RAD * T ** 4 * x ** 3 / np.expm1(x)
        assert abs(oracle.planck_function(H_PLANCK, K_B, A_RAD, nu, T) - expect) <= 1e-13 * expect
    assert oracle.planck_function(H_PLANCK, K_B, A_RAD, 1e21, T) == 0.0  # x > 100


def test_group_mean_opacity_against_quadrature(oracle):
    """ComputeGroupMeanOpacity (radiation_system.hpp:1252-1287): kappa(nu) = kappa_L (nu / nu_L)^a weighted with nu^alpha over the group"""
    from scipy.integrate import quad
    b = np.array([1.0, 3.0, 10.0, 40.0])
    expo = np.array([-2.0, 0.0, 1.5, np.nan])
    lower = np.array([5.0, 2.0, 7.0, np.nan])
    for alpha in (np.array([-1.0, -1.0, -1.0]), np.array([2.0, -4.0, 0.3]), np.array([-1.0, 1.0, -2.5])):
        k = oracle.group_mean_opacity(b, expo, lower, alpha)
        for g in range(3):
            num = quad(lambda nu: lower[g] * (nu / b[g]) ** expo[g] * nu ** alpha[g], b[g], b[g + 1])[0]
            den = quad(lambda nu: nu ** alpha[g], b[g], b[g + 1])[0]
            assert abs(k[g] - num / den) < 1e-10 * abs(num / den), (alpha, g)
    # the two guards: alpha + 1 > 100 -> value at the upper edge; < -100 -> value at the lower edge
    k = oracle.group_mean_opacity(b, expo, lower, np.array([200.0, -200.0, 0.0]))
    assert k[0] == lower[0] * (b[1] / b[0]) ** expo[0] and k[1] == lower[1]


def test_rad_quantity_exponents(oracle):
    b = np.array([1.0, 2.0, 4.0, 8.0, 16.0, 32.0])
    centre = np.sqrt(b[:-1] * b[1:])
    q = 3.0 * centre ** -1.7 * np.diff(b)  # a pure power law: the interior slopes are exact
    e = oracle.rad_quantity_exponents(b, q)
    assert e[0] == -1.0 and e[-1] == -1.0 and np.allclose(e[1:-1], -1.7, rtol=1e-13)
    # minmod: a slope change of sign gives 0; zeros on both sides give slope 0; a zero next to a positive value an "infinite" slope
    q2 = np.array([1.0, 4.0, 1.0, 0.0, 0.0])
    e2 = oracle.rad_quantity_exponents(b, q2)
    assert e2[1] == 0.0 and e2[3] == 0.0


def radshock_mg_error(U, nG=5):
    """test_radhydro_shock_multigroup.cpp:255-341: relative L1 error of T_rad / T0 (summed over the groups) against extern/LowrieEdwards/shock.txt,
    tolerance 0.008"""
    T0, Lx = 2.18e6, 0.01575
    nx = U.shape[-1]
    xs = Lx * ((np.arange(nx) + 0.5) / nx)
    Erad = sum(U[6 + 4 * g] for g in range(nG))
    Trad = np.power(Erad / A_RAD, 0.25) / T0
    ex = np.loadtxt(os.path.join(HERE, "golden", "LowrieEdwards_shock.txt"))
    m = (ex[:, 0] > 0.0) & (ex[:, 0] < Lx)
    return float(np.abs(np.interp(ex[m, 0], xs, Trad) - ex[m, 4]).sum() / np.abs(ex[m, 4]).sum())


@pytest.mark.parametrize("model,tol", [(PPL_FIXED_SLOPE, 0.008), (PIECEWISE_CONSTANT, 0.008), (PPL_FULL_SPECTRUM, 0.008)])
def test_multigroup_radiative_shock_meets_the_reference_criterion(oracle, model, tol):
    """RadhydroShockMultigroup (5 groups over 1e15..1e20 Hz, grey rho*kappa, Eddington closure, beta_order 1) on the deck's 64 cells
    (tests/radshockMG.in) to t = 1e-9 s.  The problem file selects PPL_opacity_fixed_slope_spectrum and lists the other two opacity models
    as alternatives: with a frequency-independent opacity all three describe the same physics and must meet the same tolerance."""
    s = oracle.sim(RADSHOCK_MG, 1, [64, 1, 1], [0, 0, 0], [0.01575, 1, 1], [0, 1, 1], max_grid_size=[64, 1, 1], opacity_model=model)
    assert s.ncomp == 26 and s.evolve()
    assert abs(s.time - 1.0e-9) < 1e-24
    c = s.rad_counters()
    assert c["fail_coupling"] == c["fail_outer"] == 0 and c["max_newton_iterations"] < 20
    U = s.valid(0)[:, 0, 0, :]
    err = radshock_mg_error(U)
    assert 1e-4 < err < tol, err
    # the spectrum behind the shock is the Planck spectrum of the post-shock temperature (the groups are coupled only through the gas)
    Tpost = 7.98e6
    f, _ = oracle.planck_fractions([1e15, 1e16, 1e17, 1e18, 1e19, 1e20], H_PLANCK, K_B, A_RAD, 0.0, Tpost)
    Eg = np.array([U[6 + 4 * g, -2] for g in range(5)])
    assert np.allclose(Eg[1:4] / Eg.sum(), f[1:4], rtol=0.03)


def tube_table():
    return np.loadtxt(os.path.join(HERE, "golden", "pressure_tube_initial_conditions.txt"))


# test_radiation_tube.cpp:339-364: the tabulated two-group solution the file carries (every fifth cell centre)
TUBE_E1 = [1.97806231974620e+15, 1.96003267738932e+15, 1.94139375399209e+15, 1.92211477326756e+15, 1.90216201239978e+15, 1.88149879792274e+15,
           1.86008566953792e+15, 1.83786164032564e+15, 1.81475431543605e+15, 1.79075351115540e+15, 1.76583321387339e+15, 1.73994821924481e+15,
           1.71303490596213e+15, 1.68501210246915e+15, 1.65578211109368e+15, 1.62523187429012e+15, 1.59323434543658e+15, 1.55965009717319e+15,
           1.52432919817267e+15, 1.48711344303233e+15, 1.44783897852076e+15, 1.40633941779824e+15, 1.36244954047942e+15, 1.31590909579761e+15,
           1.26631043035030e+15, 1.21323876205627e+15]
TUBE_E2 = [2.34197994225380e+15, 2.29654950261068e+15, 2.25010503500791e+15, 2.20262123173244e+15, 2.15407068960022e+15, 2.10442459607726e+15,
           2.05365387846208e+15, 2.00168645967436e+15, 1.94843446056395e+15, 1.89396262984460e+15, 1.83830481312661e+15, 1.78145970875519e+15,
           1.72339649303787e+15, 1.66406088653085e+15, 1.60338168190632e+15, 1.54127777970988e+15, 1.47766576756342e+15, 1.41246806782681e+15,
           1.34562168082733e+15, 1.27708749196767e+15, 1.20686010247924e+15, 1.13497806420176e+15, 1.06153434252058e+15, 9.86527809202386e+14,
           9.09819537649705e+14, 8.31394523943729e+14]


def test_radiation_pressure_tube_meets_the_reference_criterion(oracle):
    """RadTube (2 groups split at 3.3 T0, piecewise-constant opacity, beta_order 1): a static equilibrium between gas and radiation pressure
    must stay put for one sound-crossing time; test_radiation_tube.cpp:366-384: relative L1 of T_rad against the initial profile < 0.003.
    The per-group energies must also reproduce the solution tabulated in the file (:339-364)."""
    tab = tube_table()
    s = oracle.sim(RADTUBE, 1, [128, 1, 1], [0, 0, 0], [128.0, 1, 1], [0, 1, 1], max_grid_size=[128, 1, 1], table=[tab[:, n] for n in range(4)])
    assert s.ncomp == 14
    U0 = s.valid(0)[:, 0, 0, :].copy()
    assert s.evolve() and abs(s.time - 128.0 / 4.0295519855200705e7) < 1e-18
    U = s.valid(0)[:, 0, 0, :]
    T0 = np.power((U0[6] + U0[10]) / A_RAD, 0.25)
    T = np.power((U[6] + U[10]) / A_RAD, 0.25)
    err = float(np.abs(T - T0).sum() / np.abs(T0).sum())
    assert err < 0.003, err
    xs = np.arange(128) + 0.5
    x_exact = 0.5 + 5.0 * np.arange(26)
    e1 = np.abs(np.interp(x_exact, xs, U[6]) - TUBE_E1).sum() / np.sum(TUBE_E1)
    e2 = np.abs(np.interp(x_exact, xs, U[10]) - TUBE_E2).sum() / np.sum(TUBE_E2)
    assert e1 < 0.003 and e2 < 0.003, (e1, e2)
    assert np.abs(U[0] - U0[0]).sum() / U0[0].sum() < 0.003  # the gas has not moved either


def pulse_mg_error(Ug, Um, t):
    """test_radhydro_pulse_MG_const_kappa.cpp:344-408: T_gas, T_rad of the advected multigroup pulse, shifted back by v0 t (whole cells), and
    T_gas of the static grey pulse against T_rad of the static grey pulse; tolerance 0.006"""
    v0, mu = 2.0e8, 2.33 * 1.6605390666e-24
    nx = Ug.shape[-1]
    dx = 1024.0 / nx
    n_p = int(v0 * t / dx)
    half = int(nx / 2.0)
    shift = n_p - int((n_p + half) / nx) * nx
    cv = lambda rho: rho * K_B / (mu * (5. / 3. - 1.0))
    Trad = np.power(Ug[6] / A_RAD, 0.25)
    Tgas = Ug[5] / cv(Ug[0])
    Trad2u = np.power(sum(Um[6 + 4 * g] for g in range(4)) / A_RAD, 0.25)
    Tgas2u = Um[5] / cv(Um[0])
    idx = (np.arange(nx) - shift) % nx
    Trad2, Tgas2 = np.empty(nx), np.empty(nx)
    Trad2[idx], Tgas2[idx] = Trad2u, Tgas2u
    err = np.abs(Tgas - Trad).sum() + np.abs(Trad2 - Trad).sum() + np.abs(Tgas2 - Trad).sum()
    return float(err / (3.0 * np.abs(Trad).sum()))


def test_advected_multigroup_pulse_meets_the_reference_criterion(oracle):
    """RadhydroPulseMGconst: a grey pulse at rest and a 4-group pulse (PPL_opacity_fixed_slope_spectrum, constant kappa) advected at
    2e8 cm/s must agree after 100 steps (the file's `max_timesteps = 1e2; // for fast testing`) within 0.006."""
    geo = dict(max_grid_size=[64, 1, 1])
    g = oracle.sim(PULSE_MG_GREY, 1, [64, 1, 1], [-512.0, 0, 0], [512.0, 1, 1], [1, 1, 1], **geo)
    m = oracle.sim(PULSE_MG, 1, [64, 1, 1], [-512.0, 0, 0], [512.0, 1, 1], [1, 1, 1], **geo)
    assert g.ncomp == 10 and m.ncomp == 22
    assert g.evolve() and m.evolve() and g.istep == m.istep == 100
    for s in (g, m):
        c = s.rad_counters()
        assert c["fail_coupling"] == c["fail_outer"] == 0
    err = pulse_mg_error(g.valid(0)[:, 0, 0, :], m.valid(0)[:, 0, 0, :], m.time)
    assert err < 0.006, err


def test_marshak_wave_with_frequency_dependent_opacity_runs_clean(oracle):
    """RadMarshakVaytet (radiation only, 4 groups, kappa ~ nu^-2, PPL_opacity_full_spectrum): the reference's ctest has no error norm — it
    passes when the run reaches t_end without a Newton-Raphson failure (an abort in the reference, a counted failure here).  First tenth of
    the run here (3262 steps; tests/test_multigroup_gpu.py runs it to the end): no failure, the wave has entered the slab, the high-frequency
    groups (small opacity) run ahead of the low-frequency ones, and nothing has reached the far end."""
    s = oracle.sim(MARSHAK_VAYTET, 1, [64, 1, 1], [0.0, 0, 0], [20.0, 1, 1], [0, 1, 1], max_grid_size=[64, 1, 1], stop_time=1.36e-8)
    assert s.ncomp == 22 and s.evolve() and abs(s.time - 1.36e-8) < 1e-20
    c = s.rad_counters()
    assert c["fail_coupling"] == c["fail_outer"] == 0 and c["max_newton_iterations"] < 30
    U = s.valid(0)[:, 0, 0, :]
    Trad = np.power(sum(U[6 + 4 * g] for g in range(4)) / A_RAD, 0.25)
    assert Trad[0] > 800.0 and abs(Trad[-1] - 300.0) < 1.0 and np.all(np.diff(Trad) <= 1e-9 * Trad[:-1])
    _, E300 = oracle.planck_fractions([6e10, 6e11, 6e12, 6e13, 6e14], H_PLANCK, K_B, A_RAD, 0.0, 300.0)
    depth = [int(np.argmax(U[6 + 4 * g] < 2.0 * E300[g])) for g in range(4)]  # first cell where the group is still near its initial value
    assert depth[3] > depth[2] > depth[1], depth
    # hydro is off: the density never changes; the gas still collects the momentum the radiation deposits (x only)
    assert np.all(U[0] == 1.0e-3) and np.all(U[2:4] == 0.0) and np.all(U[1] >= 0.0) and U[1, 0] > 0.0


def raddust_error(t, u, ngroups=1):
    """test_rad_dust.cpp:172-222 (test_rad_dust_MG.cpp:186-240 sums the groups): T_gas and T_rad (= E_rad / a_rad: the problem linearises
    the emission) of cell 0 after every step against extern/data/dust/rad_dust_exact.csv (committed as data in tests/golden/), tolerance 0.0008"""
    ex = np.loadtxt(os.path.join(HERE, "golden", "rad_dust_exact.csv"), delimiter=",", skiprows=1)
    m = ex[:, 0] > 0.0
    rho = u[:, 0]
    Eint = u[:, 4] - 0.5 * (u[:, 1] ** 2 + u[:, 2] ** 2 + u[:, 3] ** 2) / rho
    Tgas = Eint / (rho * 1.0 / (1.0 * (5.0 / 3.0 - 1.0)))  # c_v = rho k_B / (mu (gamma - 1)) with k_B = mu = 1
    Trad = sum(u[:, 6 + 4 * g] for g in range(ngroups)) / 1.0
    Tg, Tr = np.interp(t, ex[m, 0], ex[m, 1]), np.interp(t, ex[m, 0], ex[m, 2])
    return float((np.abs(Tgas - Tg).sum() + np.abs(Trad - Tr).sum()) / (np.abs(Tg).sum() + np.abs(Tr).sum()))


def test_gas_dust_radiation_relaxation_meets_the_reference_criterion(oracle):
    """RadDust: the single-group exchange with the dust-gas thermal coupling model (dust temperature of Bate & Keto by a scalar Newton iteration,
    then the 2 x 2 Newton-Raphson with the dust terms in the Jacobian), 1000 steps of 1e-8 s"""
    s = oracle.sim(RADDUST, 1, [8, 1, 1], [0, 0, 0], [1.0, 1, 1], [1, 1, 1], max_grid_size=[8, 1, 1])
    t, u = s.run_record(2000, 0, (0, 0, 0))
    assert len(t) == 1000 and abs(t[-1] - 1.0e-5) < 1e-18
    c = s.rad_counters()
    assert c["fail_coupling"] == c["fail_dust"] == c["fail_outer"] == 0
    err = raddust_error(t, u)
    assert err < 0.0008, err
    # gas, dust and radiation end at the common temperature 0.6 (energy conservation: c_v T + a T = 1.5 + 0)
    assert abs(u[-1, 6] - 0.6) < 1e-5 and abs(u[-1, 5] / 1.5 - 0.6) < 1e-5


def test_multigroup_gas_dust_radiation_relaxation_meets_the_reference_criterion(oracle):
    """RadDustMG: the same relaxation with four photon groups through radiation_dust_system.hpp's coupled branch (gas, dust and every group in
    one Newton-Raphson system)"""
    s = oracle.sim(RADDUST_MG, 1, [8, 1, 1], [0, 0, 0], [1.0, 1, 1], [1, 1, 1], max_grid_size=[8, 1, 1])
    t, u = s.run_record(2000, 0, (0, 0, 0))
    assert len(t) == 1000 and abs(t[-1] - 1.0e-5) < 1e-18
    c = s.rad_counters()
    assert c["fail_coupling"] == c["fail_dust"] == c["fail_outer"] == 0 and c["decoupled"] == 0
    err = raddust_error(t, u, ngroups=4)
    assert err < 0.0008, err
    Erad = sum(u[-1, 6 + 4 * g] for g in range(4))
    assert abs(Erad - 0.6) < 1e-5 and abs(u[-1, 5] / 1.5 - 0.6) < 1e-5


def marshak_dust_error(U, t):
    """test_radiation_marshak_dust.cpp:225-268: gas temperature (stays 1), the IR group E_1 = E_L e^{-x} (t - x) and the FUV group E_2 = E_L e^{-x}
    behind the front x < t, the floor ahead of it; the first cell is skipped; tolerance 0.01"""
    n = U.shape[1]
    x = (np.arange(n) + 0.5) / n
    EL, floor = 1.0e10 * 1.0e-2 ** 4, 1.0e-10
    e2 = np.where(x < t, EL * np.exp(-x), floor)
    e1 = np.where(x < t, EL * np.exp(-x) * (t - x), floor)
    T = U[5] / 1.0  # rho = C_V = 1
    num = np.abs(T[1:] - 1.0).sum() + np.abs(U[6][1:] - e1[1:]).sum() + np.abs(U[10][1:] - e2[1:]).sum()
    return float(num / (float(n - 1) + np.abs(e1[1:]).sum() + np.abs(e2[1:]).sum()))


def test_two_group_marshak_wave_with_dust_meets_the_reference_criterion(oracle):
    """RadMarshakDust: FUV streams in from the left and is absorbed by dust that re-emits in the (optically very thick) IR group; the weak
    dust-gas coupling puts every solve on the DECOUPLED branch of radiation_dust_system.hpp (dust_model 2)"""
    s = oracle.sim(MARSHAK_DUST, 1, [256, 1, 1], [0, 0, 0], [1.0, 1, 1], [0, 1, 1], max_grid_size=[256, 1, 1])
    assert s.evolve() and abs(s.time - 0.5) < 1e-14
    c = s.rad_counters()
    assert c["fail_coupling"] == c["fail_dust"] == c["fail_outer"] == 0
    assert c["decoupled"] == c["solves"] > 0
    U = s.valid(0)[:, 0, 0, :]
    err = marshak_dust_error(U, s.time)
    assert err < 0.01, err


def line_cooling_error(t, u, heating_rate):
    """test_rad_line_cooling.cpp:228-262 (..._MG.cpp:236-303 interpolates the same curve from 1000 samples): T_gas and the energy density of the
    line group of cell 0 after every step against  C_V dT/dt = -0.1 T + heating,  E_line = the energy the gas lost to the line; tolerance 0.0005"""
    E = np.exp(-0.1 * t) * (0.1 * 1.0 - heating_rate + heating_rate * np.exp(0.1 * t)) / 0.1
    Er = -(E - 1.0 - heating_rate * t)
    T = u[:, 5] / 1.0  # rho = C_V = 1, gas at rest
    return float((np.abs(T - E).sum() + np.abs(u[:, 6] - Er).sum()) / (np.abs(E).sum() + np.abs(Er).sum()))


def test_line_cooling_and_cosmic_ray_heating_meet_the_reference_criterion(oracle):
    """RadLineCooling: one group, kappa = 0, dust model on with a vanishing coupling coefficient; the DefineNetCoolingRate /
    DefineCosmicRayHeatingRate hooks in the single-group Newton-Raphson residual and the cooled energy handed to the radiation afterwards"""
    s = oracle.sim(LINE_COOLING, 1, [8, 1, 1], [0, 0, 0], [64.0, 1, 1], [1, 1, 1], max_grid_size=[8, 1, 1], dust_coeff=1e-20)
    t, u = s.run_record(2000, 0, (0, 0, 0))
    assert len(t) == 1000 and abs(t[-1] - 10.0) < 1e-12
    c = s.rad_counters()
    assert c["fail_coupling"] == c["fail_dust"] == c["fail_outer"] == 0
    err = line_cooling_error(np.array(t), np.array(u), 0.03)
    assert err < 0.0005, err


@pytest.mark.parametrize("dust_coeff,decoupled", [(1e-20, True), (1e20, False)])
def test_multigroup_line_cooling_with_photoelectric_heating_meets_the_reference_criterion(oracle, dust_coeff, decoupled):
    """RadLineCoolingMG with both of its decks: SolveGasDustRadiationEnergyExchangeWithPE on its decoupled branch (the gas energy from the scalar
    backward-Euler solve with cooling, cosmic-ray and photoelectric terms) and on its coupled branch (the Jacobian with the extra FUV column,
    SolveLinearEqsWithLastColumn)"""
    s = oracle.sim(LINE_COOLING_MG, 1, [8, 1, 1], [0, 0, 0], [64.0, 1, 1], [1, 1, 1], max_grid_size=[8, 1, 1], dust_coeff=dust_coeff)
    t, u = s.run_record(2000, 0, (0, 0, 0))
    assert len(t) == 1000 and abs(t[-1] - 10.0) < 1e-12
    c = s.rad_counters()
    assert c["fail_coupling"] == c["fail_dust"] == c["fail_outer"] == 0
    assert c["decoupled"] == (c["solves"] if decoupled else 0)
    err = line_cooling_error(np.array(t), np.array(u), 0.03 + 0.02)
    assert err < 0.0005, err
    assert np.all(np.abs(np.array(u)[:, 6 + 4 * 3] - 1.0) < 1e-12)  # the FUV group is not absorbed (kappa = 0): its energy stays 1


def marshak_dust_pe_error(U, t):
    """test_radiation_marshak_dust_and_PE.cpp:232-262: T = 1 + (t - x) behind the FUV front, E_IR = 0, E_FUV = 1 behind / the floor ahead; the
    first cell is skipped; tolerance 0.01"""
    n = U.shape[1]
    x = (np.arange(n) + 0.5) / n
    Tex = np.where(x < t, 1.0 + 1.0 * (t - x), 1.0)
    e2 = np.where(x < t, 1.0, 1.0e-6)
    num = np.abs(U[5][1:] - Tex[1:]).sum() + np.abs(U[6][1:]).sum() + np.abs(U[10][1:] - e2[1:]).sum()
    return float(num / (np.abs(Tex[1:]).sum() + np.abs(e2[1:]).sum()))


@pytest.mark.parametrize("dust_coeff,decoupled", [(1e20, False), (1e-20, True)])
def test_photoelectric_heating_behind_a_streaming_front_meets_the_reference_criterion(oracle, dust_coeff, decoupled):
    """RadMarshakDustPE-coupled / -decoupled: transparent gas (kappa = 1e-20) heated at the rate PE_rate * E_FUV behind a free-streaming front"""
    s = oracle.sim(MARSHAK_DUST_PE, 1, [256, 1, 1], [0, 0, 0], [1.0, 1, 1], [0, 1, 1], max_grid_size=[256, 1, 1], dust_coeff=dust_coeff)
    assert s.evolve() and abs(s.time - 0.5) < 1e-14
    c = s.rad_counters()
    assert c["fail_coupling"] == c["fail_dust"] == c["fail_outer"] == 0
    assert c["decoupled"] == (c["solves"] if decoupled else 0)
    err = marshak_dust_pe_error(s.valid(0)[:, 0, 0, :], s.time)
    assert err < 0.01, err
