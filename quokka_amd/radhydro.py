"""Radiation-hydrodynamics level-0 driver over the C-ABI: the radiation half of QuokkaSimulation<problem_t>

  computeNumberOfRadiationSubsteps   reference src/QuokkaSimulation.hpp:397-406
  computeMaxSignalLocal (radhydro)   reference src/QuokkaSimulation.hpp:408-441
  subcycleRadiationAtLevel           reference src/QuokkaSimulation.hpp:1576-1722  (IMEX PD-ARS)
  advanceRadiationForwardEuler       reference src/QuokkaSimulation.hpp:1790-1821
  advanceRadiationMidpointRK2        reference src/QuokkaSimulation.hpp:1823-1857
  operatorSplitSourceTerms           reference src/QuokkaSimulation.hpp:1859-1882

State layout (Physics_Indices): comps 0..5 hydro, 6..9 = (E_r, F_x, F_y, F_z).  Host orchestration only.
Differences from the reference that do not change results: the fluxes of the old radiation state are evaluated
once per substep and reused by the midpoint stage (the reference recomputes them, :1836); prim + reconstruction +
HLL flux run as one kernel per direction; Newton counters are reduced per wave and read once per substep.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import Callable, Optional

import numpy as np
import torch

from . import capi
from .multifab import Context, MultiFab
from .simulation import Geometry, HydroSimulation

RAD0 = 6


def _p3(mfs):
    arr = (C.c_void_p * 3)()
    for d in range(3):
        arr[d] = mfs[d].ptr if d < len(mfs) else None
    return arr


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in (list(v) + [1.0, 1.0, 1.0])[:3]])


class RadhydroSimulation(HydroSimulation):
    def __init__(self, ctx: Context, geom: Geometry, traits: capi.HydroTraits, rad_traits: capi.RadTraits, bcs, max_grid_size=None,
                 rank: int = 0, nranks: int = 1, use_fused: bool = True, dirichlet=None, boxes=None, owner=None):
        self.nGroups = max(int(rad_traits.ngroups), 1)  # Physics_Traits::nGroups
        self.nrad = 4 * self.nGroups                      # Physics_NumVars::numRadVars * nGroups
        self.ncomp_override = RAD0 + self.nrad
        HydroSimulation.__init__(self, ctx, geom, traits, bcs, max_grid_size, dirichlet, rank, nranks, use_fused, ncomp_cc=RAD0 + self.nrad,
                                 boxes=boxes, owner=owner)
        self.rad_traits = rad_traits
        self.is_hydro_enabled = True  # Physics_Traits::is_hydro_enabled (False: radiation-only problems)
        self.radiationCflNumber_ = 0.3
        self.maxSubsteps_ = 10
        self.radiationReconstructionOrder_ = 3
        self.radiationCellUpdates_ = 0
        self.use_wavespeed_correction_ = False  # QuokkaSimulation.hpp:133
        self._rad_eps = None
        lev, nd = self.lev, geom.ndim
        self.radFluxOld = [MultiFab(lev, self.nrad, 0, facedir=d) for d in range(nd)]
        self.radFlux = [MultiFab(lev, self.nrad, 0, facedir=d) for d in range(nd)]
        # One transport stage as three sweeps that take the flux divergence where the fluxes are produced (qk_rad_stage_fused; one set of sweeps
        # per photon group): 3-D; the face fluxes are stored only when something reads them (store_rad_flux: the flux registers of a hierarchy)
        self.use_fused_rad = bool(use_fused) and nd == 3 and os.environ.get("QK_RAD_FUSED", "1") == "1"
        self.store_rad_flux = False
        # swapRadiationState() of substeps 2 .. n folded into the source-term kernel of the substep before (its registers hold the values)
        self.use_rad_mirror = os.environ.get("QK_RAD_MIRROR", "1") != "0"  # (the environment variable: same-box A/B, profiles/tools/ab_env.sh)
        self._rad_acc: Optional[MultiFab] = None
        self.radEnergySource = MultiFab(lev, self.nGroups, 0, fill=0.0)  # QuokkaSimulation.hpp:1866-1869
        self.SetRadEnergySource: Optional[Callable] = None  # fn(i, j, k, time) -> array on the valid box
        self._source_time_independent = True
        self._source_set = False
        # Newton counters: one slot of 4 ints per radiation substep (a slot stays below 2^31 at any box size), read once per level advance
        self.dev_rad_counter = torch.zeros(4, dtype=torch.int32, device=ctx.device)
        self._rad_counter_slot = 0
        self.dev_rad_failure = torch.zeros(3, dtype=torch.int32, device=ctx.device)
        self.rad_counters = {"solves": 0, "newton_iterations": 0, "max_newton_iterations": 0, "decoupled": 0}

    # ------------------------------------------------------------------ dt
    def computeTimestepAtLevel(self) -> float:
        if not self.is_hydro_enabled:  # QuokkaSimulation.hpp:421-424, radiation only: the signal speed is c_hat in every cell
            return self.cflNumber_ * (self.min_dx() / self.rad_traits.c_hat)
        if self._signal() is not None:
            m = self._signal_of_state_new[1]
        else:
            m = float(self.hydro.maxSignalSpeedLocal(self.lev, self.state_new_cc_, which=1, out=self.dev_max).item())
        # QuokkaSimulation.hpp:421-434: per cell max(c_hat / maxSubsteps, hydro signal); max over cells commutes
        m = max(self.rad_traits.c_hat / float(self.maxSubsteps_), m)
        m = self._allreduce_max(m)
        return self.cflNumber_ * (self.min_dx() / m)

    def computeNumberOfRadiationSubsteps(self, dt_lev_hydro: float) -> int:
        dtrad_tmp = self.radiationCflNumber_ * (self.min_dx() / self.rad_traits.c_hat)
        return int(math.ceil(dt_lev_hydro / dtrad_tmp))

    # ------------------------------------------------------------------ pieces
    def _wavespeed_eps(self, state: MultiFab):
        """use_wavespeed_correction_ (reference src/QuokkaSimulation.hpp:133, :1958-1960): the factors ComputeFluxes<DIR> puts on the dissipative part
        of the radiation-energy flux, from ComputeCellOpticalDepth<DIR> of `state` (ghost cells filled, gas components included); None when off"""
        if not self.use_wavespeed_correction_:
            return None
        if self._rad_eps is None:
            self._rad_eps = [MultiFab(self.lev, self.nGroups, 0, facedir=d) for d in range(self.geom.ndim)]
        c = self.ctx
        c.check(c.L.qk_rad_ComputeWavespeedCorrection(self.lev.h, c.stream(), C.byref(self.rad_traits), C.byref(self.traits), self.geom.ndim, state.ptr,
                                                      _d3(self.geom.dx), _p3(self._rad_eps)), "qk_rad_ComputeWavespeedCorrection")
        return _p3(self._rad_eps)

    def _rad_fluxes(self, state: MultiFab, out):
        c = self.ctx
        c.check(c.L.qk_rad_computeRadiationFluxes(self.lev.h, c.stream(), C.byref(self.rad_traits), self.geom.ndim,
                                                  self.radiationReconstructionOrder_, state.ptr, _p3(out), self._wavespeed_eps(state)),
                "qk_rad_computeRadiationFluxes")

    def _fill_source(self, time: float):
        if self.SetRadEnergySource is None or (self._source_set and self._source_time_independent):
            return
        for b, (lo, hi) in enumerate(self.my_boxes):
            k, j, i = np.meshgrid(np.arange(lo[2], hi[2] + 1), np.arange(lo[1], hi[1] + 1), np.arange(lo[0], hi[0] + 1), indexing="ij")
            src = np.ascontiguousarray(self.SetRadEnergySource(i, j, k, time))  # (nz, ny, nx) or (nGroups, nz, ny, nx)
            if src.ndim == 3:
                self.radEnergySource.fabs[b][0].copy_(torch.from_numpy(src))
            else:
                self.radEnergySource.fabs[b].copy_(torch.from_numpy(src))
        self._source_set = True

    def operatorSplitSourceTerms(self, time: float, dt: float, stage: int, mirror: bool = False):
        """mirror: the new radiation state also goes to state_old_cc_ (valid cells) — the swapRadiationState() of the next substep, stored from
        the registers of the source-term kernel (qk_rad_AddSourceTermsSingleGroupMirror)"""
        self._fill_source(time + dt)
        c = self.ctx
        if mirror:
            c.check(c.L.qk_rad_AddSourceTermsSingleGroupMirror(
                self.lev.h, c.stream(), C.byref(self.rad_traits), C.byref(self.traits), self.state_new_cc_.ptr, self.radEnergySource.ptr, float(dt), stage,
                C.c_void_p(self.dev_rad_counter.data_ptr() + 16 * self._rad_counter_slot), C.c_void_p(self.dev_rad_failure.data_ptr()),
                self.state_old_cc_.ptr), "qk_rad_AddSourceTermsSingleGroupMirror")
            return
        # QuokkaSimulation.hpp:1875-1881
        fn, name = ((c.L.qk_rad_AddSourceTermsSingleGroup, "qk_rad_AddSourceTermsSingleGroup") if self.nGroups <= 1
                    else (c.L.qk_rad_AddSourceTermsMultiGroup, "qk_rad_AddSourceTermsMultiGroup"))
        c.check(fn(self.lev.h, c.stream(), C.byref(self.rad_traits), C.byref(self.traits), self.state_new_cc_.ptr, self.radEnergySource.ptr, float(dt), stage,
                   C.c_void_p(self.dev_rad_counter.data_ptr() + 16 * self._rad_counter_slot), C.c_void_p(self.dev_rad_failure.data_ptr())), name)

    def _fill_rad_ghosts(self, state: MultiFab):
        """fillBoundaryConditions for the transport kernels, which read only the radiation components of the ghost cells: the same-rank
        copies and the physical BCs are restricted to them (strips to other ranks carry everything; the reference fills all components)"""
        L, h = self.ctx.L, self.ghost.h
        if self.use_wavespeed_correction_:  # ComputeCellOpticalDepth reads the gas state either side of every face: all components
            self.fillBoundaryConditions(state)
            return
        self.ctx.check(L.qk_ghost_plan_set_components(h, RAD0, self.nrad), "qk_ghost_plan_set_components")
        try:
            self.fillBoundaryConditions(state)
        finally:
            self.ctx.check(L.qk_ghost_plan_set_components(h, 0, -1), "qk_ghost_plan_set_components")

    def _rad_stage_fused(self, stage: int, U_in: MultiFab, U0: MultiFab, U_new: MultiFab, flux_out, dt_radiation: float):
        if self._rad_acc is None:
            self._rad_acc = MultiFab(self.lev, self.nrad, 0)
        c = self.ctx
        c.check(c.L.qk_rad_stage_fused(self.lev.h, c.stream(), C.byref(self.rad_traits), self.radiationReconstructionOrder_, stage, U_in.ptr, U0.ptr,
                                       U_new.ptr, self._rad_acc.ptr, _p3(flux_out) if self.store_rad_flux else None, float(dt_radiation),
                                       _d3(self.geom.dx), self._wavespeed_eps(U_in)), "qk_rad_stage_fused")

    def advanceRadiationForwardEuler(self, dt_radiation: float):
        self._fill_rad_ghosts(self.state_old_cc_)
        if self.use_fused_rad:
            self._rad_stage_fused(1, self.state_old_cc_, self.state_old_cc_, self.state_new_cc_, self.radFluxOld, dt_radiation)
            return
        self._rad_fluxes(self.state_old_cc_, self.radFluxOld)
        c = self.ctx
        c.check(c.L.qk_rad_PredictStep(self.lev.h, c.stream(), C.byref(self.rad_traits), self.geom.ndim, self.state_old_cc_.ptr, self.state_new_cc_.ptr,
                                       _p3(self.radFluxOld), float(dt_radiation), _d3(self.geom.dx)), "qk_rad_PredictStep")

    def advanceRadiationMidpointRK2(self, dt_radiation: float):
        self._fill_rad_ghosts(self.state_new_cc_)
        if self.use_fused_rad:  # (the Z sweep writes state_new in place: it marches every column in one thread and reads no other column)
            self._rad_stage_fused(2, self.state_new_cc_, self.state_old_cc_, self.state_new_cc_, self.radFlux, dt_radiation)
            return
        # fluxes of the old state: identical to the ones of the forward-Euler stage (state_old_cc_ and its ghosts are unchanged)
        self._rad_fluxes(self.state_new_cc_, self.radFlux)
        c = self.ctx
        c.check(c.L.qk_rad_AddFluxesRK2(self.lev.h, c.stream(), C.byref(self.rad_traits), self.geom.ndim, self.state_new_cc_.ptr, self.state_old_cc_.ptr,
                                        self.state_new_cc_.ptr, _p3(self.radFluxOld), _p3(self.radFlux), float(dt_radiation), _d3(self.geom.dx)),
                "qk_rad_AddFluxesRK2")

    def swapRadiationState(self):
        # (components are the outermost index of a fab: the radiation block of a box, ghost cells included, is one contiguous run)
        self.state_old_cc_.copy_comps_from(self.state_new_cc_, RAD0, RAD0 + self.nrad)

    def subcycleRadiationAtLevel(self, time: float, dt_lev_hydro: float) -> bool:
        if self.is_hydro_enabled and not (self.constantDt_ > 0.0):  # reference src/QuokkaSimulation.hpp:1583: radiation-only problems take ONE step
            nsub = self.computeNumberOfRadiationSubsteps(dt_lev_hydro)
            dt_rad = dt_lev_hydro / float(nsub)
        else:
            nsub, dt_rad = 1, dt_lev_hydro
        if not (1 <= nsub <= self.maxSubsteps_ + 1 and dt_rad > 0.0):
            raise capi.QkError(f"radiation substep assertion failed: nsubSteps = {nsub} (reference src/QuokkaSimulation.hpp:1596-1598)")
        self._signal_of_state_new = None  # the source terms change the gas state
        time_subcycle = time
        if self.dev_rad_counter.numel() < 4 * nsub:
            self.dev_rad_counter = torch.zeros(4 * nsub, dtype=torch.int32, device=self.ctx.device)
        c = self.ctx
        c.check(c.L.qk_clear_bytes(c.h, c.stream(), C.c_void_p(self.dev_rad_counter.data_ptr()), 4 * self.dev_rad_counter.numel()), "qk_clear_bytes")
        c.check(c.L.qk_clear_bytes(c.h, c.stream(), C.c_void_p(self.dev_rad_failure.data_ptr()), 4 * self.dev_rad_failure.numel()), "qk_clear_bytes")
        mirrored = False
        for i in range(nsub):
            if i > 0 and not mirrored:
                self.swapRadiationState()
            self._rad_counter_slot = i
            self.advanceRadiationForwardEuler(dt_rad)
            self.operatorSplitSourceTerms(time_subcycle, dt_rad, 1)  # IMEX_a22 > 0
            self.advanceRadiationMidpointRK2(dt_rad)
            mirrored = self.use_rad_mirror and self.nGroups <= 1 and i < nsub - 1
            self.operatorSplitSourceTerms(time_subcycle, dt_rad, 2, mirror=mirrored)
            time_subcycle += dt_rad
            self.radiationCellUpdates_ += self.CountCells()
        # one host read per level advance (the reference aborts inside the kernel launch that fails; here the flags of all substeps are
        # looked at together, before anything uses the state)
        fail = self._allreduce_sum_list(self.dev_rad_failure.tolist())
        cnt = self.dev_rad_counter[:4 * nsub].view(nsub, 4).to(torch.int64)
        tot, mx = cnt.sum(dim=0).tolist(), int(cnt[:, 2].max().item())
        if self.nranks > 1:  # (the C++ host reduces its counters over the ranks too: quokka_host.hpp subcycleRadiationAtLevel)
            tot, mx = self._allreduce_sum_list([int(x) for x in tot]), int(self._allreduce_max(float(mx)))
        self.rad_counters["solves"] += tot[0]
        self.rad_counters["newton_iterations"] += tot[1]
        self.rad_counters["max_newton_iterations"] = max(self.rad_counters["max_newton_iterations"], mx)
        self.rad_counters["decoupled"] += tot[3]  # multigroup dust model: solves on the decoupled gas-dust branch
        if fail[1] > 0:
            raise capi.QkError("Newton-Raphson iteration for dust temperature failed to converge or dust temperature is negative!")
        if fail[0] > 0:
            raise capi.QkError("Newton-Raphson iteration for matter-radiation coupling failed to converge!")
        if fail[2] > 0:
            raise capi.QkError("Outer iteration for matter-radiation coupling failed to converge!")
        return True

    def _allreduce_sum_list(self, vals):
        if self.nranks > 1:
            import torch.distributed as dist
            from . import comm
            t = torch.tensor(vals, dtype=torch.int64, device=self.ctx.device)
            comm.all_reduce(t, dist.ReduceOp.SUM)
            return [int(x) for x in t.tolist()]
        return vals

    # advanceSingleTimestepAtLevel (reference src/QuokkaSimulation.hpp:653-707): hydro, then the radiation subcycle
    def step(self, dt: Optional[float] = None) -> bool:
        if dt is None:
            self.computeTimestep()
        else:
            self.dt_ = dt
        time = self.tNew_
        self.tNew_ += self.dt_
        self.state_old_cc_, self.state_new_cc_ = self.state_new_cc_, self.state_old_cc_
        if self.is_hydro_enabled:
            ok = self.advanceHydroAtLevelWithRetries(self.dt_)
        else:  # QuokkaSimulation.hpp:681-685: copy hydro vars from state_old_cc_ to state_new_cc_
            self.state_new_cc_.copy_comps_from(self.state_old_cc_, 0, RAD0)
            ok = True
        if ok:
            ok = self.subcycleRadiationAtLevel(time, self.dt_)
        self.istep += 1
        self.cellUpdates_ += self.CountCells()
        return ok


# ---------------------------------------------------------------------- RadhydroShell problem generator
class ShellConstants:
    """reference src/problems/RadhydroShell/test_radhydro_shell.cpp:39-95"""
    a_rad = 7.5646e-15
    c = 2.99792458e10
    a0 = 2.0e5
    chat = 860.0 * a0
    gamma_gas = 5.0 / 3.0
    Msun = 2.0e33
    parsec_in_cm = 3.086e18
    specific_luminosity = 2000.0
    GMC_mass = 1.0e6 * Msun
    epsilon = 0.5
    M_shell = (1 - epsilon) * GMC_mass
    L_star = (epsilon * GMC_mass) * specific_luminosity
    r_0 = 5.0 * parsec_in_cm
    sigma_star = 0.3 * r_0
    H_shell = 0.3 * r_0
    kappa0 = 20.0
    rho_0 = M_shell / ((4.0 / 3.0) * math.pi * r_0 * r_0 * r_0)
    c_v = capi.K_B / ((2.2 * capi.M_U) * (gamma_gas - 1.0))
    L_box = 6.172e19  # tests/radhydro_shell_256.in


def shell_problem(ctx: Context, n: int, table, max_grid_size: int = 128, rank=0, nranks=1, pow_mode: int = 0) -> RadhydroSimulation:
    """reference src/problems/RadhydroShell/test_radhydro_shell.cpp + tests/radhydro_shell_256.in.
    `table`: (r_over_r0, Erad, Frad) columns of extern/dust_shell/initial_conditions.txt."""
    S = ShellConstants
    geom = Geometry(3, [n, n, n], [0.0, 0.0, 0.0], [S.L_box] * 3, [1, 1, 1])
    bcs = [([capi.BC_INT_DIR] * 3, [capi.BC_INT_DIR] * 3) for _ in range(10)]
    traits = capi.traits(S.gamma_gas, False, 3, mean_molecular_weight=2.2 * capi.M_U, boltzmann_constant=capi.K_B)
    rt = capi.RadTraits(S.c, S.chat, S.a_rad, 0.0, 1, 0, S.kappa0, S.kappa0, S.kappa0, pow_mode)
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, [max_grid_size] * 3, rank=rank, nranks=nranks)
    # problem_main :409-431
    for name, value in shell_settings().items():
        setattr(sim, name, value)
    ic, source = shell_functions(geom, table)
    sim.SetRadEnergySource = source
    sim.set_initial_conditions(ic)
    return sim


def shell_settings() -> dict:
    """problem_main of test_radhydro_shell.cpp (:409-431)"""
    S = ShellConstants
    return {"cflNumber_": 0.3, "densityFloor_": 1.0e-8 * S.rho_0, "reconstructionOrder_": 2, "radiationReconstructionOrder_": 2, "integratorOrder_": 2,
            "stopTime_": 0.125 * (S.r_0 / S.a0), "maxTimesteps_": 50}


def shell_functions(geom: Geometry, table):
    """(initial conditions, radiation source) of RadhydroShell on the index space of `geom` (a level of a hierarchy has its own)"""
    S = ShellConstants
    r_arr = np.asarray(table[0], dtype=np.float64) * S.r_0
    E_arr, F_arr = np.asarray(table[1], dtype=np.float64), np.asarray(table[2], dtype=np.float64)
    dx, lo, hi = geom.dx, geom.prob_lo, geom.prob_hi
    x0 = [lo[d] + 0.5 * (hi[d] - lo[d]) for d in range(3)]

    def radius(i, j, k):
        x = lo[0] + (i + 0.5) * dx[0]
        y = lo[1] + (j + 0.5) * dx[1]
        z = lo[2] + (k + 0.5) * dx[2]
        return np.sqrt((x - x0[0]) * (x - x0[0]) + (y - x0[1]) * (y - x0[1]) + (z - x0[2]) * (z - x0[2]))

    def ic(i, j, k):  # setInitialConditionsOnGrid :185-249
        r = radius(i, j, k)
        sigma_sh = S.H_shell / (2.0 * math.sqrt(2.0 * math.log(2.0)))
        rho_norm = S.M_shell / (4.0 * math.pi * r * r * math.sqrt(2.0 * math.pi * sigma_sh * sigma_sh))
        rho_shell = rho_norm * np.exp(-((r - S.r_0) * (r - S.r_0)) / (2.0 * sigma_sh * sigma_sh))
        rho = np.maximum(rho_shell, 1.0e-8 * S.rho_0)
        Frad = np.interp(r, r_arr, F_arr)
        Erad = np.interp(r, r_arr, E_arr)
        Tgas = np.power(Erad / S.a_rad, 1.0 / 4.0)
        Eint = rho * S.c_v * Tgas
        U = np.zeros((10,) + i.shape)
        U[0], U[4], U[5], U[6] = rho, Eint, Eint, Erad
        U[7] = U[8] = U[9] = Frad / math.sqrt(3.0)
        return U

    def source(i, j, k, time):  # SetRadEnergySource :98-125
        r = radius(i, j, k)
        source_norm = (1.0 / S.c) * S.L_star / math.pow(2.0 * math.pi * S.sigma_star * S.sigma_star, 1.5)
        return source_norm * np.exp(-(r * r) / (2.0 * S.sigma_star * S.sigma_star))

    return ic, source


class RadShockConstants:
    """reference src/problems/RadhydroShockCGS/test_radhydro_shock_cgs.cpp:22-56 (Skinner et al. 2019, Sec. 9.5)"""
    a_rad = 7.5646e-15
    c = 2.99792458e10
    k_B = capi.K_B
    c_s0 = 1.73e7
    kappa = 577.0  # rho * kappa [cm^-1]
    gamma_gas = 5.0 / 3.0
    m_p, m_e = 1.67262192369e-24, 9.1093837015e-28
    c_v = k_B / ((m_p + m_e) * (gamma_gas - 1.0))
    T0, rho0, v0 = 2.18e6, 5.69, 5.19e7
    T1, rho1, v1 = 7.98e6, 17.1, 1.73e7
    chat = 10.0 * (v0 + c_s0)
    Erad0 = a_rad * (T0 * T0 * T0 * T0)
    Egas0 = rho0 * c_v * T0
    Erad1 = a_rad * (T1 * T1 * T1 * T1)
    Egas1 = rho1 * c_v * T1
    shock_position = 0.01305
    Lx = 0.01575


def radshock_problem(ctx: Context, nx: int = 512, pow_mode: int = 0, beta_order: int = 1) -> RadhydroSimulation:
    """reference src/problems/RadhydroShockCGS/test_radhydro_shock_cgs.cpp + tests/radshock.in (1-D build): a steady subcritical
    radiative shock; Eddington approximation, constant absorption coefficient, constant states beyond both x faces."""
    S = RadShockConstants
    geom = Geometry(1, [nx], [0.0, 0.0, 0.0], [S.Lx, 1.0, 1.0], [0, 1, 1])
    bcs = [([capi.BC_EXT_DIR, 0, 0], [capi.BC_EXT_DIR, 0, 0]) for _ in range(10)]
    traits = capi.traits(S.gamma_gas, True, 1, mean_molecular_weight=S.m_p + S.m_e, boltzmann_constant=S.k_B)
    rt = capi.RadTraits(S.c, S.chat, S.a_rad, 0.0, beta_order, 1, S.kappa, S.kappa, S.kappa, pow_mode, 1)  # reference: beta_order 1
    pxL, pxR = S.rho0 * S.v0, S.rho1 * S.v1
    left = [S.rho0, pxL, 0.0, 0.0, S.Egas0 + (pxL * pxL) / (2 * S.rho0), S.Egas0, S.Erad0, 0.0, 0.0, 0.0]
    right = [S.rho1, pxR, 0.0, 0.0, S.Egas1 + (pxR * pxR) / (2 * S.rho1), S.Egas1, S.Erad1, 0.0, 0.0, 0.0]
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, [nx, 1, 1], use_fused=False, dirichlet={(0, 0): left, (0, 1): right})
    sim.cflNumber_ = sim.radiationCflNumber_ = 0.4  # problem_main :222-262
    sim.maxTimesteps_, sim.stopTime_ = 20000, 1.0e-9
    dx = geom.dx[0]

    def ic(i, j, k):  # setInitialConditionsOnGrid :160-219
        x = (i + 0.5) * dx
        pre = x < S.shock_position
        U = np.zeros((10,) + i.shape)
        U[0] = np.where(pre, S.rho0, S.rho1)
        U[1] = np.where(pre, S.rho0 * S.v0, S.rho1 * S.v1)
        U[4] = np.where(pre, S.Egas0 + 0.5 * S.rho0 * (S.v0 * S.v0), S.Egas1 + 0.5 * S.rho1 * (S.v1 * S.v1))
        U[5] = U[4] - (U[1] * U[1]) / (2 * U[0])
        U[6] = np.where(pre, S.Erad0, S.Erad1)
        return U

    sim.set_initial_conditions(ic)
    return sim


class StreamingConstants:
    """reference src/problems/RadStreaming/test_radiation_streaming.cpp:24-31"""
    initial_Erad = 1.0e-5
    initial_Egas = 1.0e-5
    c = 1.0
    chat = 0.2
    kappa0 = 1.0e-10
    rho = 1.0


def streaming_problem(ctx: Context, nx: int = 1000, pow_mode: int = 0) -> RadhydroSimulation:
    """reference src/problems/RadStreaming/test_radiation_streaming.cpp + tests/RadStreaming.in (1-D build): a radiation front entering
    an optically thin medium at the (reduced) speed of light; radiation only, Levermore closure at f = 1, beta_order = 0."""
    S = StreamingConstants
    geom = Geometry(1, [nx], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0, 1, 1])
    bcs = [([capi.BC_EXT_DIR, 0, 0], [capi.BC_FOEXTRAP, 0, 0]) for _ in range(10)]
    traits = capi.traits(5.0 / 3.0, True, 1, mean_molecular_weight=1.0, boltzmann_constant=1.0)
    rt = capi.RadTraits(S.c, S.chat, 1.0, S.initial_Erad, 0, 0, S.kappa0, S.kappa0, S.kappa0, pow_mode, 0)
    gas = [S.rho, 0.0, 0.0, 0.0, S.initial_Egas, S.initial_Egas]
    # setCustomBoundaryConditions :100-165 writes both ends whatever the BCRec says: the foextrap fill of the upper face is overwritten
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, [nx, 1, 1], use_fused=False,
                             dirichlet={(0, 0): gas + [1.0, S.c * 1.0, 0.0, 0.0], (0, 1): gas + [S.initial_Erad, 0.0, 0.0, 0.0]})
    sim.is_hydro_enabled = False
    sim.radiationReconstructionOrder_, sim.stopTime_, sim.radiationCflNumber_, sim.maxDt_, sim.maxTimesteps_ = 3, 1.0, 0.8, 1e-2, 5000

    def ic(i, j, k):
        U = np.zeros((10,) + i.shape)
        U[0], U[4], U[5], U[6] = S.rho, S.initial_Egas, S.initial_Egas, S.initial_Erad
        return U

    sim.set_initial_conditions(ic)
    return sim


def streaming_y_problem(ctx: Context, n_cell=(4, 100), pow_mode: int = 0, max_grid_size=None) -> RadhydroSimulation:
    """reference src/problems/RadStreamingY/test_radiation_streaming_y.cpp + tests/RadStreamingY.in (a 2-D build; c_hat = c, max_time = 0.2): the
    same front entering through the lower y face of a domain periodic in x — the radiation operators on an AMREX_SPACEDIM == 2 level."""
    S = StreamingConstants
    geom = Geometry(2, list(n_cell), [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [1, 0, 1])
    bcs = [([capi.BC_INT_DIR, capi.BC_EXT_DIR, 0], [capi.BC_INT_DIR, capi.BC_FOEXTRAP, 0]) for _ in range(10)]
    traits = capi.traits(5.0 / 3.0, True, 2, mean_molecular_weight=1.0, boltzmann_constant=1.0)
    rt = capi.RadTraits(S.c, S.c, 1.0, S.initial_Erad, 0, 0, S.kappa0, S.kappa0, S.kappa0, pow_mode, 0)
    gas = [S.rho, 0.0, 0.0, 0.0, S.initial_Egas, S.initial_Egas]
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, list(max_grid_size) if max_grid_size else [n_cell[0], n_cell[1], 1], use_fused=False,
                             dirichlet={(1, 0): gas + [1.0, 0.0, S.c * 1.0, 0.0], (1, 1): gas + [S.initial_Erad, 0.0, 0.0, 0.0]})
    sim.is_hydro_enabled = False
    sim.radiationReconstructionOrder_, sim.stopTime_, sim.radiationCflNumber_, sim.maxDt_, sim.maxTimesteps_ = 3, 0.2, 0.8, 1e-2, 5000

    def ic(i, j, k):
        U = np.zeros((10,) + i.shape)
        U[0], U[4], U[5], U[6] = S.rho, S.initial_Egas, S.initial_Egas, S.initial_Erad
        return U

    sim.set_initial_conditions(ic)
    return sim


class SuOlsonConstants:
    """reference src/problems/RadSuOlson/test_radiation_SuOlson.cpp:21-33"""
    eps_SuOlson, kappa, rho0, T_hohlraum, x0, t0 = 1.0, 1.0, 1.0, 1.0, 0.5, 10.0
    a_rad, c = 1.0, 1.0
    alpha_SuOlson = 4.0 * a_rad / eps_SuOlson
    Q = 1.0 / (2.0 * x0)
    S = Q * (a_rad * (T_hohlraum * T_hohlraum * T_hohlraum * T_hohlraum))


def suolson_problem(ctx: Context, nx: int = 1500, pow_mode: int = 0) -> RadhydroSimulation:
    """reference src/problems/RadSuOlson/test_radiation_SuOlson.cpp + tests/SuOlson.in (1-D build): a radiation source switched on in
    x < x0 heats a cold half-space with heat capacity alpha T^3 (the material of Su & Olson 1997: E = alpha / 4 T^4); radiation only,
    reflecting walls, kappa = 1 / rho, beta_order 0."""
    S = SuOlsonConstants
    geom = Geometry(1, [nx], [0.0, 0.0, 0.0], [30.0, 1.0, 1.0], [0, 0, 0])
    bcs = []
    for n in range(10):
        odd = n in (1, RAD0 + 1)
        bcs.append(([capi.BC_REFLECT_ODD if odd else capi.BC_REFLECT_EVEN, 0, 0], [capi.BC_REFLECT_ODD if odd else capi.BC_REFLECT_EVEN, 0, 0]))
    traits = capi.traits(5.0 / 3.0, True, 1, mean_molecular_weight=1.0, boltzmann_constant=1.0, eos_temperature_model=1, eos_alpha=S.alpha_SuOlson)
    rt = capi.RadTraits(S.c, S.c, S.a_rad, 0.0, 0, 1, S.kappa, S.kappa, S.kappa, pow_mode, 0)
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, [nx, 1, 1], use_fused=False)
    sim.is_hydro_enabled = False
    sim.cflNumber_ = sim.radiationCflNumber_ = 0.4
    sim.stopTime_, sim.maxTimesteps_, sim.maxDt_, sim.initDt_ = 10.0, 12000, 1e-2, 1e-9
    dx = geom.dx[0]
    initial_Egas = 1e-10 * ((S.alpha_SuOlson / 4.0) * ((S.T_hohlraum * S.T_hohlraum) * (S.T_hohlraum * S.T_hohlraum)))
    initial_Erad = 1e-10 * (S.a_rad * (S.T_hohlraum * S.T_hohlraum * S.T_hohlraum * S.T_hohlraum))

    def ic(i, j, k):
        U = np.zeros((10,) + i.shape)
        U[0], U[4], U[5], U[6] = S.rho0, initial_Egas, initial_Egas, initial_Erad
        return U

    def source(i, j, k, time):  # SetRadEnergySource :115-145
        xl, xr = (i + 0.0) * dx, (i + 1.0) * dx
        frac = np.where((xl < S.x0) & (xr <= S.x0), 1.0, np.where((xl < S.x0) & (xr > S.x0), (S.x0 - xl) / (xr - xl), 0.0))
        return (S.S * frac) if time < S.t0 else 0.0 * frac

    sim.SetRadEnergySource = source
    sim._source_time_independent = False
    sim.set_initial_conditions(ic)
    return sim


class CouplingConstants:
    """reference src/problems/RadMatterCoupling/test_radiation_matter_coupling.cpp:24-27, :113-115"""
    a_rad = 7.5646e-15
    alpha_SuOlson = 4.0 * a_rad / 1.0
    Erad0, Egas0, rho0 = 1.0e12, 1.0e2, 1.0e-7
    c_light = 2.99792458e10
    a_rad_cgs = 4.0 * 5.670374419e-5 / c_light  # C::a_rad: the radiation constant the solver uses


def matter_coupling_problem(ctx: Context, n: int = 4, pow_mode: int = 0, c_hat_factor: float = 1.0) -> RadhydroSimulation:
    """reference src/problems/RadMatterCoupling/test_radiation_matter_coupling.cpp + tests/energyexchange.in (1-D build): gas and
    radiation of a uniform medium relax to a common temperature; constant dt = 1e-8 s, kappa = 1, E_gas = alpha / 4 T^4."""
    S = CouplingConstants
    geom = Geometry(1, [n], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0, 0, 0])
    bcs = [([capi.BC_FOEXTRAP, 0, 0], [capi.BC_FOEXTRAP, 0, 0]) for _ in range(10)]
    traits = capi.traits(5.0 / 3.0, True, 1, eos_temperature_model=1, eos_alpha=S.alpha_SuOlson)
    # c_hat_factor = 0.1: RadMatterCouplingRSLA (test_radiation_matter_coupling_rsla.cpp:22,43), the same problem with a reduced speed of light
    rt = capi.RadTraits(S.c_light, c_hat_factor * S.c_light, S.a_rad_cgs, 0.0, 1, 0, 1.0, 1.0, 1.0, pow_mode, 0)
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, [n, 1, 1], use_fused=False)
    sim.is_hydro_enabled = False
    sim.cflNumber_ = sim.radiationCflNumber_ = 1.0
    sim.constantDt_, sim.maxTimesteps_, sim.stopTime_ = 1.0e-8, 1000000, 1.0e-2

    def ic(i, j, k):
        U = np.zeros((10,) + i.shape)
        U[0], U[4], U[5], U[6] = S.rho0, S.Egas0, S.Egas0, S.Erad0
        return U

    sim.set_initial_conditions(ic)
    return sim


class AdvectingConstants:
    """reference src/problems/RadhydroUniformAdvecting/test_radhydro_uniform_advecting.cpp:44-57 ("model 3")"""
    c = 1.0e8
    chat = 1.0e8
    v0 = 1e-2 * c
    kappa0 = 1.0e5
    T0 = rho0 = a_rad = mu = k_B = 1.0
    max_time = 10.0 / v0
    Erad0 = a_rad * T0 * T0 * T0 * T0
    Erad_beta2 = (1.0 + 4.0 / 3.0 * (v0 * v0) / (c * c)) * Erad0


def uniform_advecting_problem(ctx: Context, nx: int = 64, pow_mode: int = 0) -> RadhydroSimulation:
    """reference src/problems/RadhydroUniformAdvecting/test_radhydro_uniform_advecting.cpp + tests/RadhydroUniformAdvecting.in (1-D
    build): gas and radiation in equilibrium advect at 0.01 c through a periodic box with beta_order = 2; T_gas must stay at T0."""
    S = AdvectingConstants
    geom = Geometry(1, [nx], [0.0, 0.0, 0.0], [64.0, 1.0, 1.0], [1, 1, 1])
    bcs = [([capi.BC_INT_DIR, 0, 0], [capi.BC_INT_DIR, 0, 0]) for _ in range(10)]
    traits = capi.traits(5.0 / 3.0, True, 1, mean_molecular_weight=S.mu, boltzmann_constant=S.k_B)
    rt = capi.RadTraits(S.c, S.chat, S.a_rad, 0.0, 2, 0, S.kappa0, S.kappa0, S.kappa0, pow_mode, 0)
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, [nx, 1, 1], use_fused=False)
    sim.radiationReconstructionOrder_ = 3  # problem_main :139-165
    sim.stopTime_, sim.radiationCflNumber_, sim.cflNumber_, sim.maxDt_, sim.maxTimesteps_ = S.max_time, 8.0, 0.8, 1.0, 1000000
    # quokka::EOS::ComputeEintFromTgas (EOS.hpp:116-159) with the gamma-law network, in its order of operations (the CGS constants cancel
    # up to rounding, which the bit-for-bit comparison sees)
    mu_ = S.mu / capi.M_U
    pres = S.rho0 * S.T0 * capi.K_B / (mu_ * capi.M_U)
    Egas = pres / ((5.0 / 3.0 - 1.0) * S.rho0) * S.rho0 * S.k_B / capi.K_B

    def ic(i, j, k):  # setInitialConditionsOnGrid :84-126, the beta_order_ == 2 branch
        U = np.zeros((10,) + i.shape)
        U[0], U[1] = S.rho0, S.v0 * S.rho0
        U[4], U[5] = Egas + 0.5 * S.rho0 * S.v0 * S.v0, Egas
        U[6], U[7] = S.Erad_beta2, 4.0 / 3.0 * S.v0 * S.Erad0
        return U

    sim.set_initial_conditions(ic)
    return sim


class MarshakConstants:
    """reference src/problems/RadMarshak/test_radiation_marshak.cpp:22-31"""
    eps_SuOlson = 1.0
    kappa = 1.0
    rho0 = 1.0
    T_hohlraum = 1.0
    a_rad = 1.0
    c = 1.0
    alpha_SuOlson = 4.0 * a_rad / eps_SuOlson
    T_initial = 1.0e-2


def marshak_problem(ctx: Context, nx: int = 80, pow_mode: int = 0) -> RadhydroSimulation:
    """reference src/problems/RadMarshak/test_radiation_marshak.cpp + tests/Marshak.in (1-D build): the Su & Olson (1996) Marshak wave:
    a half-space of the E = alpha / 4 T^4 material lit through its lower face by a hohlraum at T_H (Marshak half-range condition on
    the ghost flux), constant state beyond the upper face; radiation only, kappa = 1, beta_order 0."""
    S = MarshakConstants
    geom = Geometry(1, [nx], [0.0, 0.0, 0.0], [20.0, 1.0, 1.0], [0, 1, 1])
    bcs = [([capi.BC_EXT_DIR, 0, 0], [capi.BC_FOEXTRAP, 0, 0]) for _ in range(10)]
    traits = capi.traits(5.0 / 3.0, True, 1, mean_molecular_weight=1.0, boltzmann_constant=1.0, eos_temperature_model=1, eos_alpha=S.alpha_SuOlson)
    rt = capi.RadTraits(S.c, S.c, S.a_rad, 0.0, 0, 0, S.kappa, S.kappa, S.kappa, pow_mode, 0)
    Egas = (S.alpha_SuOlson / 4.0) * ((S.T_initial * S.T_initial) * (S.T_initial * S.T_initial))  # quokka::EOS::ComputeEintFromTgas hook :73-79
    Erad = S.a_rad * math.pow(S.T_initial, 4)
    E_inc = S.a_rad * math.pow(S.T_hohlraum, 4)
    gas = [S.rho0, 0.0, 0.0, 0.0, Egas, Egas]
    # setCustomBoundaryConditions :101-160 (it does not consult the side's BCRec: the cells beyond the upper face get the constant state)
    dirichlet = {(0, 0): {"values": gas + [E_inc, 0.0, 0.0, 0.0], "marshak": (RAD0, RAD0 + 1, S.c)}, (0, 1): gas + [Erad, 0.0, 0.0, 0.0]}
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, [nx, 1, 1], use_fused=False, dirichlet=dirichlet)
    sim.is_hydro_enabled = False
    chi = S.rho0 * S.kappa  # problem_main :181-218
    sim.radiationCflNumber_ = 0.4
    sim.stopTime_ = 10.0 / (S.eps_SuOlson * S.c * chi)
    sim.maxDt_ = 1e-3 / (S.eps_SuOlson * S.c * chi)
    sim.initDt_ = 1e-9 / (S.eps_SuOlson * S.c * chi)
    sim.maxTimesteps_ = 20000

    def ic(i, j, k):  # setInitialConditionsOnGrid :162-183
        U = np.zeros((10,) + i.shape)
        U[0], U[4], U[5], U[6] = S.rho0, Egas, Egas, Erad
        return U

    sim.set_initial_conditions(ic)
    return sim


class RadForceConstants:
    """reference src/problems/RadForce/test_radiation_force.cpp:33-46"""
    c_light = 2.99792458e10
    a_rad = 4.0 * 5.670374419e-5 / c_light
    kappa0 = 5.0
    mu = 2.33 * capi.M_U
    a0 = 0.2e5
    tau = 1.0e-6
    rho0 = 1.0e5 * mu
    Mach0 = 1.1
    Mach1 = 2.128410288469465339
    Frad0 = rho0 * a0 * c_light / tau
    g0 = kappa0 * Frad0 / c_light
    Lx = (a0 * a0) / g0


def radforce_problem(ctx: Context, nx: int = 128, pow_mode: int = 0) -> RadhydroSimulation:
    """reference src/problems/RadForce/test_radiation_force.cpp + tests/RadForce.in (1-D build): an isothermal gas (gamma = 1,
    cs_isothermal = a0) entering at Mach 1.1 is accelerated by the force of an optically thin radiation flux (Planck opacity 0,
    flux-mean opacity 5) to the steady wind solution; inflow state beyond the lower face, extrapolation beyond the upper one."""
    S = RadForceConstants
    geom = Geometry(1, [nx], [0.0, 0.0, 0.0], [1.0263747986171498e16, 1.0, 1.0], [0, 1, 1])  # tests/RadForce.in
    bcs = [([capi.BC_EXT_DIR, 0, 0], [capi.BC_FOEXTRAP, 0, 0]) for _ in range(10)]
    traits = capi.traits(1.0, True, 1, mean_molecular_weight=S.mu, boltzmann_constant=capi.K_B, cs_isothermal=S.a0)
    rt = capi.RadTraits(S.c_light, 10.0 * (S.Mach1 * S.a0), S.a_rad, 0.0, 1, 0, 0.0, 0.0, S.kappa0, pow_mode, 0)
    inflow = [S.rho0, S.rho0 * (S.Mach0 * S.a0), 0.0, 0.0, 0.0, 0.0, S.Frad0 / S.c_light, S.Frad0, 0.0, 0.0]
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, [nx, 1, 1], use_fused=False, dirichlet={(0, 0): inflow})
    sim.radiationReconstructionOrder_ = 3  # problem_main :166-205
    sim.reconstructionOrder_ = 3
    sim.stopTime_ = 10.0 * (S.Lx / S.a0)
    sim.cflNumber_ = sim.radiationCflNumber_ = 0.4
    sim.maxTimesteps_, sim.maxDt_ = 1000000, 1.0e10

    def ic(i, j, k):  # setInitialConditionsOnGrid :84-114
        U = np.zeros((10,) + i.shape)
        U[0] = S.rho0
        U[6], U[7] = S.Frad0 * 1.0 / S.c_light, S.Frad0 * 1.0
        return U

    sim.set_initial_conditions(ic)
    return sim


class MarshakAsymptoticConstants:
    """reference src/problems/RadMarshakAsymptotic/test_radiation_marshak_asymptotic.cpp:19-28"""
    kappa = 300.0
    rho0 = 2.0879373766122384
    T_hohlraum = 1.1604448449e7
    T_initial = T_hohlraum * 0.001
    c_light = 2.99792458e10
    a_rad = 4.0 * 5.670374419e-5 / c_light
    Erad_floor = a_rad * T_initial * T_initial * T_initial * T_initial


def marshak_asymptotic_problem(ctx: Context, nx: int = 60, pow_mode: int = 0) -> RadhydroSimulation:
    """reference src/problems/RadMarshakAsymptotic/test_radiation_marshak_asymptotic.cpp + tests/MarshakAsymptotic.in (1-D build): a
    Marshak wave in the equilibrium-diffusion limit (McClarren & Lowrie 2008): gamma-law gas, absorption coefficient
    300 (T / T_H)^-3 cm^-1 (`opacity_model = 2`), Eddington approximation, Marshak condition on the lower face, radiation only."""
    S = MarshakAsymptoticConstants
    geom = Geometry(1, [nx], [0.0, 0.0, 0.0], [0.66, 1.0, 1.0], [0, 1, 1])
    bcs = [([capi.BC_EXT_DIR, 0, 0], [capi.BC_FOEXTRAP, 0, 0]) for _ in range(10)]
    traits = capi.traits(5.0 / 3.0, True, 1, mean_molecular_weight=capi.M_U, boltzmann_constant=capi.K_B)
    rt = capi.RadTraits(S.c_light, S.c_light, S.a_rad, S.Erad_floor, 0, 2, S.kappa, S.kappa, S.kappa, pow_mode, 1, S.T_hohlraum, -3.0, 0.0)
    # quokka::EOS::ComputeEintFromTgas (EOS.hpp:116-159) in its order of operations
    mu_ = capi.M_U / capi.M_U
    pres = S.rho0 * S.T_initial * capi.K_B / (mu_ * capi.M_U)
    Egas = pres / ((5.0 / 3.0 - 1.0) * S.rho0) * S.rho0 * capi.K_B / capi.K_B
    Erad = S.a_rad * math.pow(S.T_initial, 4)
    E_inc = S.a_rad * math.pow(S.T_hohlraum, 4)
    gas = [S.rho0, 0.0, 0.0, 0.0, Egas, Egas]
    dirichlet = {(0, 0): {"values": gas + [E_inc, 0.0, 0.0, 0.0], "marshak": (RAD0, RAD0 + 1, S.c_light)}, (0, 1): gas + [Erad, 0.0, 0.0, 0.0]}
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, [nx, 1, 1], use_fused=False, dirichlet=dirichlet)
    sim.is_hydro_enabled = False
    sim.radiationReconstructionOrder_ = 3  # problem_main :200-250
    sim.stopTime_, sim.initDt_, sim.maxDt_, sim.radiationCflNumber_, sim.maxTimesteps_ = 10.0e-9, 5.0e-12, 5.0, 10.0, 1000000

    def ic(i, j, k):  # setInitialConditionsOnGrid :142-164
        U = np.zeros((10,) + i.shape)
        U[0], U[4], U[5], U[6] = S.rho0, Egas, Egas, Erad
        return U

    sim.set_initial_conditions(ic)
    return sim


class RadPulseConstants:
    """reference src/problems/RadPulse/test_radiation_pulse.cpp:20-29"""
    kappa0, T0, rho0, a_rad, c, chat = 1.0e5, 1.0, 1.0, 4.0e-10, 1.0e8, 1.0e7
    erad_floor = a_rad * (1.0e-10)
    initial_time = 1.0e-8


def radpulse_exact_Trad(x, t):
    """compute_exact_Trad (test_radiation_pulse.cpp:58-68): the diffusion solution for the Gaussian pulse"""
    S = RadPulseConstants
    sigma = 0.025
    D = 4.0 * S.c * S.a_rad * math.pow(S.T0, 3) / (3.0 * S.kappa0)
    width_sq = sigma * sigma + D * t
    normfac = 1.0 / (2.0 * np.sqrt(np.pi * width_sq))
    return 0.5 * normfac * np.exp(-(x * x) / (4.0 * width_sq))


def radpulse_problem(ctx: Context, nx: int = 32, pow_mode: int = 0, initial_state=None) -> RadhydroSimulation:
    """reference src/problems/RadPulse/test_radiation_pulse.cpp + tests/RadPulse.in (1-D build): a Gaussian temperature pulse diffusing
    in a medium with kappa = (kappa0 / rho) max((T / T0)^3, 1) — `opacity_model = 2` with exponent 3 and floor 1 —, optical depth
    ~1e5 per cell at the peak; radiation only, c_hat = c / 10, extrapolation at both faces.  `initial_state` (10, 1, 1, nx): start from
    a given state instead of evaluating the Gaussian here (numpy's exp and libm's may differ in the last place)."""
    S = RadPulseConstants
    geom = Geometry(1, [nx], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0, 1, 1])
    bcs = [([capi.BC_FOEXTRAP, 0, 0], [capi.BC_FOEXTRAP, 0, 0]) for _ in range(10)]
    traits = capi.traits(5.0 / 3.0, True, 1, mean_molecular_weight=1.0, boltzmann_constant=2.0 / 3.0)
    rt = capi.RadTraits(S.c, S.chat, S.a_rad, S.erad_floor, 0, 2, S.kappa0, S.kappa0, S.kappa0, pow_mode, 0, S.T0, 3.0, 1.0)
    sim = RadhydroSimulation(ctx, geom, traits, rt, bcs, [nx, 1, 1], use_fused=False)
    sim.is_hydro_enabled = False
    sim.radiationReconstructionOrder_ = 3  # problem_main :121-150
    sim.stopTime_, sim.radiationCflNumber_, sim.maxDt_, sim.maxTimesteps_ = 1.0e-4, 0.8, 1e-3, 100000
    dx = geom.dx[0]

    def ic(i, j, k):  # setInitialConditionsOnGrid :81-108
        if initial_state is not None:
            return np.asarray(initial_state)[:, k, j, i]
        x = (i + 0.5) * dx
        Trad = radpulse_exact_Trad(x - 0.5, S.initial_time)
        mu_ = 1.0 / capi.M_U  # quokka::EOS::ComputeEintFromTgas in its order of operations
        pres = S.rho0 * Trad * capi.K_B / (mu_ * capi.M_U)
        Egas = pres / ((5.0 / 3.0 - 1.0) * S.rho0) * S.rho0 * (2.0 / 3.0) / capi.K_B
        U = np.zeros((10,) + i.shape)
        U[0], U[4], U[5], U[6] = S.rho0, Egas, Egas, S.erad_floor
        return U

    sim.set_initial_conditions(ic)
    return sim
