"""Problem generators for the multigroup radiation path (RadhydroSimulation with rad_traits.ngroups > 1): the reference's multigroup test
problems on the HIP operators.  Planck energy fractions in initial and boundary states come from the library's own device function
(qk_rad_mg_planck_fractions), never from the CPU oracle.

  src/problems/RadhydroShockMultigroup/test_radhydro_shock_multigroup.cpp   radshock_mg_problem
  src/problems/RadTube/test_radiation_tube.cpp                              radtube_problem
  src/problems/RadMarshakVaytet/test_radiation_marshak_Vaytet.cpp           marshak_vaytet_problem
  src/problems/RadhydroPulseMGconst/test_radhydro_pulse_MG_const_kappa.cpp  pulse_mg_problem
  src/problems/RadDust/test_rad_dust.cpp (single group + dust)              raddust_problem
"""
import ctypes as C
import math

import numpy as np

from . import capi
from .multifab import Context
from .radhydro import RAD0, RadhydroSimulation
from .simulation import Geometry

PIECEWISE_CONSTANT, PPL_FIXED_SLOPE, PPL_FULL_SPECTRUM = 1, 2, 3  # OpacityModel (radiation_system.hpp:64-71)

C_LIGHT = 2.99792458e10
A_RAD = 4.0 * 5.670374419e-5 / C_LIGHT
H_PLANCK = 6.62607015e-27
M_P, M_E = 1.67262192369e-24, 9.1093837015e-28


def planck_fractions(ctx: Context, rt: capi.RadTraits, kB: float, T):
    """ComputePlanckEnergyFractions / ComputeThermalRadiationMultiGroup of the library at the temperatures T -> (fractions, E_g), each (n, nGroups)"""
    T = np.ascontiguousarray(np.atleast_1d(T), dtype=np.float64).ravel()
    ng = rt.ngroups
    f, E = np.empty((T.size, ng)), np.empty((T.size, ng))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    ctx.check(ctx.L.qk_rad_mg_planck_fractions(ctx.h, C.byref(rt), float(kB), int(T.size), dp(T), dp(f), dp(E)), "qk_rad_mg_planck_fractions")
    return f, E


def eint_from_tgas(rho, T, mu, kB=capi.K_B, gamma=5.0 / 3.0):
    """quokka::EOS::ComputeEintFromTgas (gamma law) in its order of operations (reference src/hydro/EOS.hpp:109-141)"""
    mu_ = mu / capi.M_U
    pres = rho * T * capi.K_B / (mu_ * capi.M_U)
    return pres / ((gamma - 1.0) * rho) * rho * kB / capi.K_B


def rad_state(ncomp, shape):
    return np.zeros((ncomp,) + tuple(shape))


# ---------------------------------------------------------------------- RadhydroShockMultigroup
class RadShockMGConstants:
    """test_radhydro_shock_multigroup.cpp:19-43"""
    a_rad, c, k_B = A_RAD, C_LIGHT, capi.K_B
    c_s0, kappa, gamma_gas = 1.73e7, 577.0, 5.0 / 3.0
    c_v = k_B / ((M_P + M_E) * (gamma_gas - 1.0))
    T0, rho0, v0 = 2.18e6, 5.69, 5.19e7
    T1, rho1, v1 = 7.98e6, 17.1, 1.73e7
    chat = 10.0 * (v0 + c_s0)
    Erad0 = a_rad * (T0 * T0 * T0 * T0)
    Erad_floor = Erad0 * 1e-12
    Egas0, Egas1 = rho0 * c_v * T0, rho1 * c_v * T1
    shock_position, Lx = 0.0130, 0.01575
    boundaries = [1.0e15, 1.0e16, 1.0e17, 1.0e18, 1.0e19, 1.0e20]


def radshock_mg_problem(ctx: Context, nx: int = 64, opacity_model: int = PPL_FIXED_SLOPE, pow_mode: int = 0, three_d: bool = False,
                        max_grid_size=None, nyz: int = 4) -> RadhydroSimulation:
    """5 photon groups over 1e15..1e20 Hz, grey absorption coefficient 577 cm^-1 (exponent 0, lower value 577 / rho), Eddington closure,
    constant states beyond both x faces (deck tests/radshockMG.in: 64 cells; three_d: the deck's own 64 x 4 x 4 cells, periodic in y and z)."""
    S = RadShockMGConstants
    ng = len(S.boundaries) - 1
    ncomp = RAD0 + 4 * ng
    if three_d:
        geom = Geometry(3, [nx, nyz, nyz], [0.0, 0.0, 0.0], [S.Lx, 0.001575, 1.0], [0, 1, 1])  # (nyz > 4: kernel timing, profiles/tools/mg_kernel_time.py)
        mgs = max_grid_size if max_grid_size is not None else [nx, nyz, nyz]
    else:
        geom = Geometry(1, [nx], [0.0, 0.0, 0.0], [S.Lx, 1.0, 1.0], [0, 1, 1])
        mgs = [nx, 1, 1]
    bcs = [([capi.BC_EXT_DIR, capi.BC_INT_DIR, capi.BC_INT_DIR], [capi.BC_EXT_DIR, capi.BC_INT_DIR, capi.BC_INT_DIR]) for _ in range(ncomp)]
    traits = capi.traits(S.gamma_gas, True, geom.ndim, mean_molecular_weight=M_P + M_E, boltzmann_constant=S.k_B)
    rt = capi.RadTraits(S.c, S.chat, S.a_rad, S.Erad_floor, 1, 0, 0.0, 0.0, 0.0, pow_mode, 1)
    rt.set_groups(S.boundaries, H_PLANCK, opacity_model, [0.0] * (ng + 1), [S.kappa] * (ng + 1), rho_exponent=-1.0)
    _, Eg = planck_fractions(ctx, rt, S.k_B, [S.T0, S.T1])

    def side(rho, v, Egas, E):
        px = rho * v
        out = [rho, px, 0.0, 0.0, Egas + (px * px) / (2 * rho), Egas]
        for g in range(ng):
            out += [E[g], 0.0, 0.0, 0.0]
        return out

    dirichlet = {(0, 0): side(S.rho0, S.v0, S.Egas0, Eg[0]), (0, 1): side(S.rho1, S.v1, S.Egas1, Eg[1])}
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, mgs, use_fused=False, dirichlet=dirichlet)
    sim.cflNumber_ = sim.radiationCflNumber_ = 0.4  # problem_main :221-247
    sim.maxTimesteps_, sim.stopTime_ = 20000, 1.0e-9
    dx = geom.dx[0]

    def ic(i, j, k):  # setInitialConditionsOnGrid :164-219
        x = (i + 0.5) * dx
        pre = x < S.shock_position
        U = rad_state(ncomp, i.shape)
        U[0] = np.where(pre, S.rho0, S.rho1)
        U[1] = np.where(pre, S.rho0 * S.v0, S.rho1 * S.v1)
        U[4] = np.where(pre, S.Egas0 + 0.5 * S.rho0 * (S.v0 * S.v0), S.Egas1 + 0.5 * S.rho1 * (S.v1 * S.v1))
        U[5] = U[4] - (U[1] * U[1]) / (2 * U[0])
        for g in range(ng):
            U[RAD0 + 4 * g] = np.where(pre, Eg[0][g], Eg[1][g])
        return U

    sim.set_initial_conditions(ic)
    return sim


# ---------------------------------------------------------------------- RadTube
class RadTubeConstants:
    """test_radiation_tube.cpp:30-41"""
    kappa0, mu, gamma_gas = 100.0, 2.33 * capi.M_U, 5.0 / 3.0
    rho0, T0 = 1.0, 2.75e7
    rho1, T1 = 2.1940476649492044, 2.2609633884436745e7
    a_rad, a0, Lx = A_RAD, 4.0295519855200705e7, 128.0
    boundaries = [0.01 * 2.75e7, 3.3 * 2.75e7, 1000.0 * 2.75e7]  # Kelvin (energy_unit = k_B)


def radtube_problem(ctx: Context, table, nx: int = 128, pow_mode: int = 0) -> RadhydroSimulation:
    """A static balance of gas and radiation pressure, 2 groups split at 3.3 T0, piecewise-constant opacity 100 cm^2/g.  `table`: the columns
    (x, rho, Pgas, Erad) of extern/pressure_tube/initial_conditions.txt.  Beyond the x faces: constant density / temperature, the normal
    momentum and the normal radiation fluxes follow the first valid cell (qk_dirichlet_face::interior_mask)."""
    S = RadTubeConstants
    ng = 2
    ncomp = RAD0 + 4 * ng
    geom = Geometry(1, [nx], [0.0, 0.0, 0.0], [S.Lx, 1.0, 1.0], [0, 1, 1])
    bcs = [([capi.BC_EXT_DIR, 0, 0], [capi.BC_EXT_DIR, 0, 0]) for _ in range(ncomp)]
    traits = capi.traits(S.gamma_gas, True, 1, mean_molecular_weight=S.mu, boltzmann_constant=capi.K_B)
    rt = capi.RadTraits(C_LIGHT, 10.0 * S.a0, S.a_rad, 0.0, 1, 0, 0.0, 0.0, 0.0, pow_mode, 0)
    rt.set_groups(S.boundaries, capi.K_B, PIECEWISE_CONSTANT, [0.0] * (ng + 1), [S.kappa0] * (ng + 1))
    fB, _ = planck_fractions(ctx, rt, capi.K_B, [S.T0, S.T1])

    def side(rhoB, TB, frac):
        Erad = S.a_rad * math.pow(TB, 4)
        Egas = (capi.K_B / S.mu) * rhoB * TB / (S.gamma_gas - 1.0)
        out = [rhoB, 0.0, 0.0, 0.0, Egas, Egas]
        for g in range(ng):
            out += [Erad * frac[g], 0.0, 0.0, 0.0]
        return {"values": out, "interior": [1] + [RAD0 + 4 * g + 1 for g in range(ng)], "kinetic_from_interior": True}

    dirichlet = {(0, 0): side(S.rho0, S.T0, fB[0]), (0, 1): side(S.rho1, S.T1, fB[1])}
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, [nx, 1, 1], use_fused=False, dirichlet=dirichlet)
    sim.cflNumber_ = sim.radiationCflNumber_ = 0.4  # problem_main :254-285
    sim.stopTime_, sim.maxTimesteps_ = S.Lx / S.a0, 2000
    x_arr, rho_arr, P_arr, E_arr = (np.asarray(c, dtype=np.float64) for c in table)
    dx = geom.dx[0]

    def ic(i, j, k):  # setInitialConditionsOnGrid :139-182 (interpolate_value: linear between the table's points)
        x = (i + 0.5) * dx
        rho, Pgas, Erad = np.interp(x, x_arr, rho_arr), np.interp(x, x_arr, P_arr), np.interp(x, x_arr, E_arr)
        Tgas = Pgas / capi.K_B * S.mu / rho
        frac, _ = planck_fractions(ctx, rt, capi.K_B, Tgas)
        U = rad_state(ncomp, i.shape)
        for g in range(ng):
            U[RAD0 + 4 * g] = Erad * frac[:, g].reshape(i.shape)
        U[0], U[4], U[5] = rho, Pgas / (S.gamma_gas - 1.0), Pgas / (S.gamma_gas - 1.0)
        return U

    sim.set_initial_conditions(ic)
    return sim


# ---------------------------------------------------------------------- RadMarshakVaytet
class MarshakVaytetConstants:
    """test_radiation_marshak_Vaytet.cpp:23-94 (the_model = 10, n_groups_ = 4)"""
    kappa0, nu_pivot = 2000.0, 4.0e13
    rho0, T_initial, T_L, T_R = 1.0e-3, 300.0, 1000.0, 300.0
    rho_C_V = 1.0e-3
    c_v = rho_C_V / rho0
    mu = 1.0 / (5.0 / 3.0 - 1.0) * capi.K_B / c_v
    a_rad = A_RAD
    Erad_floor = a_rad * T_initial * T_initial * T_initial * T_initial * 1e-20
    boundaries = [6.0e10, 6.0e11, 6.0e12, 6.0e13, 6.0e14]


def marshak_vaytet_problem(ctx: Context, nx: int = 64, opacity_model: int = PPL_FULL_SPECTRUM, pow_mode: int = 0) -> RadhydroSimulation:
    """Radiation only, 4 groups, kappa(nu) = 2000 (nu / 4e13 Hz)^-2 cm^2/g (exponent -2 in every group, lower value at the group's lower edge;
    piecewise_constant_opacity: value at the bin centre), a 1000 K source beyond the lower face, the 300 K state beyond the upper one
    (deck tests/MarshakVaytet.in: cfl = 0.4, 64 cells on 20 cm)."""
    S = MarshakVaytetConstants
    ng = 4
    ncomp = RAD0 + 4 * ng
    geom = Geometry(1, [nx], [0.0, 0.0, 0.0], [20.0, 1.0, 1.0], [0, 1, 1])
    bcs = [([capi.BC_EXT_DIR, 0, 0], [capi.BC_FOEXTRAP, 0, 0]) for _ in range(ncomp)]
    traits = capi.traits(5.0 / 3.0, True, 1, mean_molecular_weight=S.mu, boltzmann_constant=capi.K_B)
    rt = capi.RadTraits(C_LIGHT, C_LIGHT, S.a_rad, S.Erad_floor, 0, 0, 0.0, 0.0, 0.0, pow_mode, 0)
    b = S.boundaries
    if opacity_model == PIECEWISE_CONSTANT:  # :150-155
        lower = [S.kappa0 * math.pow(math.sqrt(b[g] * b[g + 1]) / S.nu_pivot, -2.0) for g in range(ng)] + [0.0]
    else:  # :156-160
        lower = [S.kappa0 * math.pow(b[g] / S.nu_pivot, -2.0) for g in range(ng + 1)]
    rt.set_groups(b, H_PLANCK, opacity_model, [-2.0] * (ng + 1), lower)
    _, Eg = planck_fractions(ctx, rt, capi.K_B, [S.T_L, S.T_R, S.T_initial])
    Egas = eint_from_tgas(S.rho0, S.T_initial, S.mu)
    # setCustomBoundaryConditions :167-219 fills BOTH sides (the functor runs on every cell outside the domain and does not consult the
    # BCRec: the foextrap record of the upper face is overwritten by the 300 K state)
    left, right = [S.rho0, 0.0, 0.0, 0.0, Egas, Egas], [S.rho0, 0.0, 0.0, 0.0, Egas, Egas]
    for g in range(ng):
        left += [Eg[0][g], 0.0, 0.0, 0.0]
        right += [Eg[1][g], 0.0, 0.0, 0.0]
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, [nx, 1, 1], use_fused=False, dirichlet={(0, 0): left, (0, 1): right})
    sim.is_hydro_enabled = False
    sim.radiationReconstructionOrder_ = 3  # problem_main :255-285
    sim.stopTime_, sim.maxDt_, sim.radiationCflNumber_, sim.cflNumber_, sim.maxTimesteps_ = 1.36e-7, 1.0, 0.8, 0.4, 1000000

    def ic(i, j, k):  # setInitialConditionsOnGrid :221-251
        U = rad_state(ncomp, i.shape)
        U[0], U[4], U[5] = S.rho0, Egas, Egas
        for g in range(ng):
            U[RAD0 + 4 * g] = Eg[2][g]
        return U

    sim.set_initial_conditions(ic)
    return sim


# ---------------------------------------------------------------------- RadhydroPulseMGconst
class PulseMGConstants:
    """test_radhydro_pulse_MG_const_kappa.cpp:24-66"""
    kappa0, T0, T1, rho0 = 100.0, 1.0e7, 2.0e7, 1.2
    a_rad, c = A_RAD, C_LIGHT
    chat, width = C_LIGHT, 24.0
    Erad0 = A_RAD * 1.0e7 * 1.0e7 * 1.0e7 * 1.0e7
    erad_floor = Erad0 * 1.0e-14
    mu = 2.33 * capi.M_U
    v0, max_time, max_timesteps = 2.0e8, 4.8e-5, 100
    boundaries = [1e15, 1e16, 1e17, 1e18, 1e19]


def pulse_mg_problem(ctx: Context, multigroup: bool, nx: int = 64, pow_mode: int = 0) -> RadhydroSimulation:
    """multigroup = False: problem 1 of the file (grey, gas at rest); True: problem 2 (4 groups, PPL_opacity_fixed_slope_spectrum, constant
    kappa, advected at v0).  Deck tests/RadhydroPulse.in: 64 cells on [-512, 512] cm, periodic."""
    S = PulseMGConstants
    ng = 4 if multigroup else 1
    ncomp = RAD0 + 4 * ng
    geom = Geometry(1, [nx], [-512.0, 0.0, 0.0], [512.0, 1.0, 1.0], [1, 1, 1])
    bcs = [([capi.BC_INT_DIR, 0, 0], [capi.BC_INT_DIR, 0, 0]) for _ in range(ncomp)]
    traits = capi.traits(5.0 / 3.0, True, 1, mean_molecular_weight=S.mu, boltzmann_constant=capi.K_B)
    rt = capi.RadTraits(S.c, S.chat, S.a_rad, S.erad_floor, 1, 0, S.kappa0, S.kappa0, S.kappa0, pow_mode, 0)
    if multigroup:
        rt.set_groups(S.boundaries, H_PLANCK, PPL_FIXED_SLOPE, [0.0] * (ng + 1), [S.kappa0] * (ng + 1))
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, [nx, 1, 1], use_fused=False)
    sim.radiationReconstructionOrder_ = 3  # problem_main :252-280 / :311-335
    sim.stopTime_, sim.radiationCflNumber_, sim.cflNumber_, sim.maxDt_, sim.maxTimesteps_ = S.max_time, 0.8, 0.8, 1e-3, S.max_timesteps
    dx = geom.dx[0]

    def ic(i, j, k):  # :122-148, :196-232
        x = -512.0 + (i + 0.5) * dx
        T = S.T0 + (S.T1 - S.T0) * np.exp(-x * x / (2.0 * S.width * S.width))
        rho = S.rho0 * S.T0 / T + (S.a_rad * S.mu / 3.0 / capi.K_B) * (np.power(S.T0, 4) / T - np.power(T, 3))
        Egas = eint_from_tgas(rho, T, S.mu)
        U = rad_state(ncomp, i.shape)
        if multigroup:
            _, Eg = planck_fractions(ctx, rt, capi.K_B, T)
            for g in range(ng):
                E = Eg[:, g].reshape(i.shape)
                U[RAD0 + 4 * g], U[RAD0 + 4 * g + 1] = E, 4.0 / 3.0 * S.v0 * E
            U[0], U[1], U[4], U[5] = rho, S.v0 * rho, Egas + 0.5 * rho * S.v0 * S.v0, Egas
        else:
            U[RAD0] = S.a_rad * T * T * T * T
            U[0], U[4], U[5] = rho, Egas, Egas
        return U

    sim.set_initial_conditions(ic)
    return sim


# ---------------------------------------------------------------------- RadDust
class RadDustConstants:
    """test_rad_dust.cpp:19-34"""
    c = chat = 1.0e8
    chi0, T0, rho0, a_rad, mu, k_B = 10000.0, 1.0, 1.0, 1.0, 1.0, 1.0
    max_time, delta_time = 1.0e-5, 1.0e-8
    erad_floor = 1.0e-20 * (a_rad * T0 * T0 * T0 * T0)
    dust_gas_interaction_coeff = 1.0e6  # tests/RadDust.in


def raddust_problem(ctx: Context, nx: int = 8, multigroup: bool = False) -> RadhydroSimulation:
    """Gas, dust and radiation of a uniform medium relaxing to a common temperature: ISM_Traits::enable_dust_gas_thermal_coupling_model, emission
    linear in T_dust (the problem's ComputeThermalRadiationSingleGroup hook), kappa = chi0 / rho; constant dt = 1e-8 s, 1000 steps
    (deck tests/RadDust.in: 8 cells, periodic, radiation.cfl = 8, dust_gas_interaction_coeff = 1e6).  multigroup: RadDustMG
    (test_rad_dust_MG.cpp: 4 groups with edges 1e-3, 0.1, 1, 10, 1e3 in units of k_B T = 1, PPL_opacity_fixed_slope_spectrum, the floor in every group)."""
    S = RadDustConstants
    ng = 4 if multigroup else 1
    ncomp = RAD0 + 4 * ng
    geom = Geometry(1, [nx], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [1, 1, 1])
    bcs = [([capi.BC_INT_DIR, 0, 0], [capi.BC_INT_DIR, 0, 0]) for _ in range(ncomp)]
    traits = capi.traits(5.0 / 3.0, True, 1, mean_molecular_weight=S.mu, boltzmann_constant=S.k_B)
    rt = capi.RadTraits(S.c, S.chat, S.a_rad, S.erad_floor, 1, 1, S.chi0, S.chi0, S.chi0, 0, 0)
    rt.enable_dust_gas_thermal_coupling_model, rt.dust_gas_interaction_coeff, rt.thermal_model = 1, S.dust_gas_interaction_coeff, 1
    rt.gas_dust_coupling_threshold = 1.0e-6  # ISM_Traits default (radiation_system.hpp:89)
    if multigroup:  # test_rad_dust_MG.cpp:52-81
        rt.set_groups([1.0e-3, 0.1, 1.0, 10.0, 1.0e3], 1.0, PPL_FIXED_SLOPE, [0.0] * (ng + 1), [S.chi0] * (ng + 1), rho_exponent=-1.0)
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, [nx, 1, 1], use_fused=False)
    sim.radiationReconstructionOrder_ = 3  # problem_main :145-170
    sim.stopTime_, sim.cflNumber_, sim.radiationCflNumber_, sim.maxTimesteps_ = S.max_time, 0.8, 8.0, 1000000
    sim.initDt_ = sim.maxDt_ = S.delta_time
    Egas = eint_from_tgas(S.rho0, S.T0, S.mu, kB=S.k_B)

    def ic(i, j, k):  # setInitialConditionsOnGrid :99-122 (MG :106-129)
        U = rad_state(ncomp, i.shape)
        U[0], U[4], U[5] = S.rho0, Egas, Egas
        for g in range(ng):
            U[RAD0 + 4 * g] = S.erad_floor
        return U

    sim.set_initial_conditions(ic)
    return sim


# ---------------------------------------------------------------------- RadMarshakDust
class MarshakDustConstants:
    """test_radiation_marshak_dust.cpp:19-37 and the deck tests/RadMarshakDust.in"""
    c = chat = 1.0
    rho0, CV, initial_T = 1.0, 1.0, 1.0
    mu = 1.5 / CV
    a_rad, erad_floor, initial_Trad, T_rad_L = 1.0e10, 1.0e-10, 1.0e-5, 1.0e-2
    EradL = a_rad * T_rad_L * T_rad_L * T_rad_L * T_rad_L
    kappa1, kappa2 = 1.0e10, 1.0
    dust_gas_interaction_coeff, gas_dust_coupling_threshold = 1.0e-2, 1.0e-5
    boundaries = [1e-10, 100.0, 1e4]


def marshak_dust_problem(ctx: Context, nx: int = 256, pow_mode: int = 0) -> RadhydroSimulation:
    """FUV radiation (group 2, kappa = 1) streaming in through the lower face, absorbed by dust that re-emits it in the IR (group 1, kappa = 1e10):
    radiation only, beta_order 0, the weak gas-dust exchange puts every cell on the decoupled branch of radiation_dust_system.hpp.  The gas
    state is written on every cell outside the domain, the radiation state only beyond the lower face (setCustomBoundaryConditions :126-176):
    beyond the extrapolating upper face the radiation components follow the first cell inside."""
    S = MarshakDustConstants
    ng = 2
    ncomp = RAD0 + 4 * ng
    geom = Geometry(1, [nx], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0, 1, 1])
    bcs = [([capi.BC_EXT_DIR, 0, 0], [capi.BC_FOEXTRAP, 0, 0]) for _ in range(ncomp)]
    traits = capi.traits(5.0 / 3.0, True, 1, mean_molecular_weight=S.mu, boltzmann_constant=1.0)
    rt = capi.RadTraits(S.c, S.chat, S.a_rad, S.erad_floor, 0, 0, 0.0, 0.0, 0.0, pow_mode, 0)
    rt.enable_dust_gas_thermal_coupling_model, rt.dust_gas_interaction_coeff = 1, S.dust_gas_interaction_coeff
    rt.gas_dust_coupling_threshold = S.gas_dust_coupling_threshold
    rt.set_groups(S.boundaries, 1.0, PIECEWISE_CONSTANT, [0.0] * (ng + 1), [S.kappa1, S.kappa2, S.kappa2])  # :86-102
    Egas = S.initial_T * S.CV
    gas = [S.rho0, 0.0, 0.0, 0.0, Egas, Egas]
    left = gas + [S.erad_floor, S.erad_floor * S.c, 0.0, 0.0, S.EradL, S.EradL * S.c, 0.0, 0.0]
    right = {"values": gas + [0.0] * (4 * ng), "interior": list(range(RAD0, ncomp))}
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, [nx, 1, 1], use_fused=False, dirichlet={(0, 0): left, (0, 1): right})
    sim.is_hydro_enabled = False
    sim.radiationReconstructionOrder_ = 3  # problem_main :186-212
    sim.stopTime_, sim.maxDt_, sim.radiationCflNumber_, sim.maxTimesteps_ = 0.5, 1.0, 0.8, 5000
    _, Eg = planck_fractions(ctx, rt, 1.0, [S.initial_Trad])

    def ic(i, j, k):  # setInitialConditionsOnGrid :104-124
        U = rad_state(ncomp, i.shape)
        U[0], U[4], U[5] = S.rho0, Egas, Egas
        for g in range(ng):
            U[RAD0 + 4 * g] = Eg[0][g]
        return U

    sim.set_initial_conditions(ic)
    return sim


# ---------------------------------------------------------------------- RadLineCooling / RadLineCoolingMG
class LineCoolingConstants:
    """test_rad_line_cooling.cpp:19-36, test_rad_line_cooling_MG.cpp:19-42"""
    cooling_rate, CR_heating_rate, PE_rate = 0.1, 0.03, 0.02
    c = chat = 1.0
    kappa0, T0, rho0, a_rad, mu, k_B, nu_unit = 0.0, 1.0, 1.0, 1.0, 1.5, 1.0, 1.0
    erad_floor = a_rad * 1e-20
    Erad_FUV = a_rad * T0 * T0 * T0 * T0
    max_time, the_dt, line_index = 10.0, 1.0e-2, 0
    boundaries = [1.00000000e-03, 1.77827941e-02, 3.16227766e-01, 5.62341325e+00, 1.00000000e+02]


def line_cooling_problem(ctx: Context, multigroup: bool, dust_coeff: float, nx: int = 8) -> RadhydroSimulation:
    """A uniform transparent medium (kappa = 0) cooled by a line (rate 0.1 T into group 0), heated by cosmic rays (0.03) and — four groups —
    photoelectrically by the FUV group (0.02 E_FUV): the ISM hooks of the dust exchange.  dust_coeff: radiation.dust_gas_interaction_coeff of
    the deck (tests/RadLineCooling.in: 1e-20, the decoupled branch; RadLineCoolingCoupled.in: 1e20)."""
    S = LineCoolingConstants
    ng = 4 if multigroup else 1
    ncomp = RAD0 + 4 * ng
    geom = Geometry(1, [nx], [0.0, 0.0, 0.0], [64.0, 1.0, 1.0], [1, 1, 1])
    bcs = [([capi.BC_INT_DIR, 0, 0], [capi.BC_INT_DIR, 0, 0]) for _ in range(ncomp)]
    traits = capi.traits(5.0 / 3.0, True, 1, mean_molecular_weight=S.mu, boltzmann_constant=S.k_B)
    rt = capi.RadTraits(S.c, S.chat, S.a_rad, S.erad_floor, 0, 0, S.kappa0, S.kappa0, S.kappa0, 0, 0)
    rt.enable_dust_gas_thermal_coupling_model, rt.dust_gas_interaction_coeff, rt.gas_dust_coupling_threshold = 1, float(dust_coeff), 1.0e-6
    rt.cooling_linear_coeff[S.line_index] = S.cooling_rate
    rt.cr_heating_rate = S.CR_heating_rate
    if multigroup:
        rt.set_groups(S.boundaries, S.nu_unit, PIECEWISE_CONSTANT, [0.0] * (ng + 1), [S.kappa0] * (ng + 1))
        rt.enable_photoelectric_heating, rt.pe_heating_E1_derivative = 1, S.PE_rate / S.Erad_FUV
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, [nx, 1, 1], use_fused=False)
    sim.radiationReconstructionOrder_ = 3
    sim.stopTime_, sim.cflNumber_, sim.radiationCflNumber_, sim.maxTimesteps_ = S.max_time, 0.8, 0.8, 1000000
    sim.initDt_ = sim.maxDt_ = S.the_dt
    Egas = eint_from_tgas(S.rho0, S.T0, S.mu, kB=S.k_B)

    def ic(i, j, k):
        U = rad_state(ncomp, i.shape)
        U[0], U[4], U[5] = S.rho0, Egas, Egas
        for g in range(ng):
            U[RAD0 + 4 * g] = S.Erad_FUV if (multigroup and g == ng - 1) else S.erad_floor
        return U

    sim.set_initial_conditions(ic)
    return sim


# ---------------------------------------------------------------------- RadMarshakDustPE
class MarshakDustPEConstants:
    """test_radiation_marshak_dust_and_PE.cpp:19-36 and the decks tests/RadMarshakDustPE{coupled,decoupled}.in"""
    PE_rate = 1.0
    c = chat = 1.0
    rho0, CV, initial_T, a_rad, erad_floor, T_rad_L = 1.0, 1.0, 1.0, 1.0, 1.0e-6, 1.0
    mu = 1.5 / CV
    EradL = a_rad * T_rad_L * T_rad_L * T_rad_L * T_rad_L
    kappa1 = kappa2 = 1e-20
    boundaries = [1e-10, 30.0, 1e4]


def marshak_dust_pe_problem(ctx: Context, dust_coeff: float, nx: int = 256) -> RadhydroSimulation:
    """FUV radiation streaming freely (kappa = 1e-20) through gas it heats photoelectrically at the rate PE_rate * E_FUV; dust_coeff 1e20 / 1e-20
    selects the coupled / decoupled branch of SolveGasDustRadiationEnergyExchangeWithPE.  Boundary functor as RadMarshakDust's."""
    S = MarshakDustPEConstants
    ng = 2
    ncomp = RAD0 + 4 * ng
    geom = Geometry(1, [nx], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0, 1, 1])
    bcs = [([capi.BC_EXT_DIR, 0, 0], [capi.BC_FOEXTRAP, 0, 0]) for _ in range(ncomp)]
    traits = capi.traits(5.0 / 3.0, True, 1, mean_molecular_weight=S.mu, boltzmann_constant=1.0)
    rt = capi.RadTraits(S.c, S.chat, S.a_rad, S.erad_floor, 1, 0, 0.0, 0.0, 0.0, 0, 0)
    rt.enable_dust_gas_thermal_coupling_model, rt.dust_gas_interaction_coeff, rt.gas_dust_coupling_threshold = 1, float(dust_coeff), 1.0e-4
    rt.enable_photoelectric_heating, rt.pe_heating_E1_derivative = 1, S.PE_rate
    rt.set_groups(S.boundaries, 1.0, PIECEWISE_CONSTANT, [0.0] * (ng + 1), [S.kappa1, S.kappa2, S.kappa2])
    Egas = S.initial_T * S.CV
    gas = [S.rho0, 0.0, 0.0, 0.0, Egas, Egas]
    left = gas + [S.erad_floor, S.erad_floor * S.c, 0.0, 0.0, S.EradL, S.EradL * S.c, 0.0, 0.0]
    right = {"values": gas + [0.0] * (4 * ng), "interior": list(range(RAD0, ncomp))}
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, [nx, 1, 1], use_fused=False, dirichlet={(0, 0): left, (0, 1): right})
    sim.is_hydro_enabled = False
    sim.radiationReconstructionOrder_ = 3
    sim.stopTime_, sim.maxDt_, sim.radiationCflNumber_, sim.maxTimesteps_ = 0.5, 1.0, 0.8, 5000

    def ic(i, j, k):
        U = rad_state(ncomp, i.shape)
        U[0], U[4], U[5] = S.rho0, Egas, Egas
        for g in range(ng):
            U[RAD0 + 4 * g] = S.erad_floor
        return U

    sim.set_initial_conditions(ic)
    return sim
