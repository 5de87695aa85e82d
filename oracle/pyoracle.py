"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes/numpy binding of oracle/liboracle.so (the CPU restatement of the reference algorithm).
Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
Arrays are Fortran-order with the component index outermost (amrex::Array4 layout), i.e. a numpy
array of shape (ncomp, nz, ny, nx) in C order.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

K_B = 1.380649e-16
M_U = 1.6605390666e-24

SOD, CONTACT, SEDOV, SHELL, RADSHOCK, STREAMING, SCALARS, HYDRO1D, COUPLING, SUOLSON, ADVECTING, MARSHAK, RADFORCE, MARSHAK_ASYMPTOTIC, RADPULSE, SHOCKTUBE_CMA = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15
# multigroup radiation (oracle/problems_multigroup.hpp); PULSE_MG = the advecting 4-group run, PULSE_MG_GREY = the static grey run of the same file
RADSHOCK_MG, RADTUBE, MARSHAK_VAYTET, PULSE_MG, PULSE_MG_GREY, RADDUST = 16, 17, 18, 19, 20, 21
QUIRK = 22  # HydroQuirk: the 2-D (or 3-D) odd-even decoupling test
RADDUST_MG = 24  # RadDustMG: the same relaxation with 4 photon groups (multigroup dust exchange)
LINE_COOLING = 26  # RadLineCooling: one group, line cooling linear in T + cosmic-ray heating (dust_coeff = the deck's coefficient)
LINE_COOLING_MG = 27  # RadLineCoolingMG: four groups + photoelectric heating by the last group
MARSHAK_DUST_PE = 28  # RadMarshakDustPE: FUV front heating the gas photoelectrically
STREAMING_Y = 29  # RadStreamingY: the streaming front along y in a 2-D build
ADVECTION_SAWTOOTH, ADVECTION_SEMIELLIPSE, ADVECTION_SQUARE_2D = 31, 32, 33  # linear advection of a scalar (AdvectionSimulation)
GENERAL_OPACITY = 30  # kappa = kappa0 rho^0.3 T^-1.7 (no reference problem: the pin of the compiled opacity hooks of the C++ host)
MARSHAK_DUST = 25  # RadMarshakDust: two groups (IR / FUV), dust model with the decoupled branch
BLAST2D = 23  # HydroBlast2D: circular blast in a reflecting box, as a 2-D build or as a 3-D build uniform in z
# OpacityModel (radiation_system.hpp:64-71)
PIECEWISE_CONSTANT, PPL_FIXED_SLOPE, PPL_FULL_SPECTRUM = 1, 2, 3


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (seconds)."""
    lib = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(lib):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


class HydroTraits(C.Structure):
    _fields_ = [
        ("gamma", C.c_double),
        ("cs_isothermal", C.c_double),
        ("mean_molecular_weight", C.c_double),
        ("boltzmann_constant", C.c_double),
        ("reconstruct_eint", C.c_int),
        ("nscalars", C.c_int),
        ("ndim", C.c_int),
    ]


def traits(gamma=1.4, reconstruct_eint=True, ndim=3, nscalars=0, mean_molecular_weight=M_U, boltzmann_constant=K_B,
           cs_isothermal=float("nan")) -> HydroTraits:
    return HydroTraits(gamma, cs_isothermal, mean_molecular_weight, boltzmann_constant, int(reconstruct_eint), nscalars, ndim)


class SimConfig(C.Structure):
    _fields_ = [
        ("problem", C.c_int),
        ("ndim", C.c_int),
        ("n_cell", C.c_int * 3),
        ("max_grid_size", C.c_int * 3),
        ("prob_lo", C.c_double * 3),
        ("prob_hi", C.c_double * 3),
        ("periodic", C.c_int * 3),
        ("cfl", C.c_double),
        ("stop_time", C.c_double),
        ("max_timesteps", C.c_long),
        ("reconstruction_order", C.c_int),
        ("nscalars", C.c_int),
        ("table_len", C.c_int),
        ("table_r", C.POINTER(C.c_double)),
        ("table_Erad", C.POINTER(C.c_double)),
        ("table_Frad", C.POINTER(C.c_double)),
        ("rad_pow_mode", C.c_int),
        ("h1d", C.c_double * 12),
        ("h1d_i", C.c_int * 2),
        ("table_extra", C.POINTER(C.c_double)),
        ("opacity_model", C.c_int),
    ]


_I3 = C.c_int * 3


def _i3(v):
    return _I3(*[int(x) for x in v])


def _dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def usable_cores() -> int:
    """threads worth using: the affinity mask, capped by twice the cgroup CPU quota (the GPU boxes show 256 CPUs and grant 16)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(round(2 * float(quota) / float(period)))))
    except (OSError, ValueError):
        pass
    if "OMP_NUM_THREADS" in os.environ:
        n = int(os.environ["OMP_NUM_THREADS"])
    return n


class Oracle:
    def __init__(self, variant: str = "direct"):
        build()
        name = "liboracle.so" if variant == "direct" else "liboracle_eosT.so"
        self.lib = C.CDLL(os.path.join(_HERE, name))
        L = self.lib
        L.orc_sim_create.restype = C.c_void_p
        L.orc_sim_create.argtypes = [C.POINTER(SimConfig)]
        for f in ("orc_sim_destroy",):
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("orc_sim_nboxes", "orc_sim_ncomp", "orc_sim_nghost", "orc_sim_step", "orc_sim_evolve"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_int
        for f in ("orc_sim_time", "orc_sim_dt", "orc_sim_compute_dt"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_double
        for f in ("orc_sim_istep", "orc_sim_cell_updates"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_long
        L.orc_sim_box.argtypes = [C.c_void_p, C.c_int, _I3, _I3]
        L.orc_sim_get_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.orc_sim_set_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.orc_sim_fill_ghosts.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.orc_sim_counters.argtypes = [C.c_void_p, C.c_long * 3]
        L.orc_sim_advance_fixed_dt.argtypes = [C.c_void_p, C.c_double]
        L.orc_sim_advance_fixed_dt.restype = C.c_int
        L.orc_eos_variant.restype = C.c_int
        L.orc_sim_rad_counters.argtypes = [C.c_void_p, C.c_long * 8]
        L.orc_sim_tag_relative_gradient.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p]
        L.orc_sim_tag_centered_gradient.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p]
        L.orc_sim_rad_source.argtypes = [C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_double)]
        L.orc_sim_rad_transport_only.argtypes = [C.c_void_p, C.c_double]
        D, DP = C.c_double, C.POINTER(C.c_double)
        L.orc_planck_integral.argtypes, L.orc_planck_integral.restype = [D], D
        L.orc_planck_table_entry.argtypes, L.orc_planck_table_entry.restype = [C.c_int], D
        L.orc_planck_fractions.argtypes = [C.c_int, DP, D, D, D, D, D, DP, DP]
        L.orc_planck_function.argtypes, L.orc_planck_function.restype = [D, D, D, D, D], D
        L.orc_group_mean_opacity.argtypes = [C.c_int, DP, DP, DP, DP, DP]
        L.orc_rad_quantity_exponents.argtypes = [C.c_int, DP, DP, DP]

    # ---------------------------------------------------------------- multigroup helper functions
    def planck_integral(self, x: float) -> float:
        return self.lib.orc_planck_integral(float(x))

    def planck_table(self) -> np.ndarray:
        return np.array([self.lib.orc_planck_table_entry(j) for j in range(1000)])

    def planck_fractions(self, boundaries, energy_unit, k_B, a_rad, Erad_floor, T):
        b = np.ascontiguousarray(boundaries, dtype=np.float64)
        n = len(b) - 1
        f, E = np.empty(n), np.empty(n)
        self.lib.orc_planck_fractions(n, _dp(b), energy_unit, k_B, a_rad, Erad_floor, T, _dp(f), _dp(E))
        return f, E

    def planck_function(self, energy_unit, k_B, a_rad, nu, T) -> float:
        return self.lib.orc_planck_function(energy_unit, k_B, a_rad, nu, T)

    def group_mean_opacity(self, boundaries, expo, lower, alpha_quant) -> np.ndarray:
        b, e, l, a = (np.ascontiguousarray(v, dtype=np.float64) for v in (boundaries, expo, lower, alpha_quant))
        k = np.empty(len(b) - 1)
        self.lib.orc_group_mean_opacity(len(b) - 1, _dp(b), _dp(e), _dp(l), _dp(a), _dp(k))
        return k

    def rad_quantity_exponents(self, boundaries, quant) -> np.ndarray:
        b, q = (np.ascontiguousarray(v, dtype=np.float64) for v in (boundaries, quant))
        e = np.empty(len(q))
        self.lib.orc_rad_quantity_exponents(len(q), _dp(b), _dp(q), _dp(e))
        return e

    # ---------------------------------------------------------------- per-operator
    def cons_to_prim(self, t: HydroTraits, cons: np.ndarray, glo, ghi) -> np.ndarray:
        prim = np.empty_like(cons)
        self.lib.orc_cons_to_prim(C.byref(t), _dp(cons), _dp(prim), _i3(glo), _i3(ghi))
        return prim

    def flattening_coefficients(self, t: HydroTraits, d: int, prim: np.ndarray, glo, ghi, clo, chi) -> np.ndarray:
        shape = tuple(int(chi[a] - clo[a] + 1) for a in (2, 1, 0))
        out = np.empty(shape, dtype=np.float64)
        self.lib.orc_flattening_coefficients(C.byref(t), int(d), _dp(prim), _i3(glo), _i3(ghi), _dp(out), _i3(clo), _i3(chi))
        return out

    def compute_hydro_fluxes(self, t: HydroTraits, order: int, cons: np.ndarray, vlo, vhi, nghost=4, K_visc=0.0, mhd_stub=False):
        """cons: (nvar, ...) on the valid box grown by nghost. Returns ([flux_d], [facevel_d]).  mhd_stub: Physics_Traits::is_mhd_enabled — the
        interface fluxes come from HLLD with the reference's B = 0 stub (hydro_system.hpp:987-1003) instead of HLLC"""
        if mhd_stub:
            assert order in (1, 2, 3)
            order = order + 10
        nv = 6 + t.nscalars
        fl, fv = [], []
        for d in range(3):
            if d < t.ndim:
                n = [int(vhi[a] - vlo[a] + 1) for a in range(3)]
                n[d] += 1
                fl.append(np.empty((nv, n[2], n[1], n[0]), dtype=np.float64))
                fv.append(np.empty((n[2], n[1], n[0]), dtype=np.float64))
            else:
                fl.append(np.empty((0,), dtype=np.float64))
                fv.append(np.empty((0,), dtype=np.float64))
        self.lib.orc_compute_hydro_fluxes(C.byref(t), int(order), C.c_double(K_visc), _dp(cons), _i3(vlo), _i3(vhi), int(nghost),
                                          _dp(fl[0]), _dp(fl[1]), _dp(fl[2]), _dp(fv[0]), _dp(fv[1]), _dp(fv[2]))
        return fl[: t.ndim], fv[: t.ndim]

    # ---------------------------------------------------------------- whole simulation
    def average_down(self, fine: np.ndarray, flo, crse: np.ndarray, clo, region, scomp: int, ncomp: int, ratio=(2, 2, 2)):
        """amrex::average_down restated: fine / crse are (ncomp_total, nz, ny, nx) float64 arrays whose lower corners are flo / clo;
        region = (lo, hi) in coarse indices; crse is modified in place"""
        assert fine.dtype == np.float64 and crse.dtype == np.float64 and fine.flags["C_CONTIGUOUS"] and crse.flags["C_CONTIGUOUS"]
        fhi = [flo[d] + fine.shape[3 - d] - 1 for d in range(3)]
        chi = [clo[d] + crse.shape[3 - d] - 1 for d in range(3)]
        L = self.lib
        L.orc_average_down.argtypes = [C.c_void_p, _I3, _I3, C.c_void_p, _I3, _I3, C.c_int, _I3, _I3, C.c_int, C.c_int, _I3]
        L.orc_average_down(fine.ctypes.data_as(C.c_void_p), _i3(flo), _i3(fhi), crse.ctypes.data_as(C.c_void_p), _i3(clo), _i3(chi), fine.shape[0],
                           _i3(region[0]), _i3(region[1]), scomp, ncomp, _i3(ratio))

    def interp_from_coarse(self, fine: np.ndarray, flo, crse_old: np.ndarray, crse_new: np.ndarray, clo, region, w_old: float, w_new: float, ncomp: int,
                           method: int = 1, hooks: bool = True, ndim: int = 3, ratio=(2, 2, 2)):
        """coarse -> fine interpolation restated (oracle/amr.hpp): fills `region` = (lo, hi) (fine indices) of `fine` in place"""
        for a in (fine, crse_old, crse_new):
            assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
        fhi = [flo[d] + fine.shape[3 - d] - 1 for d in range(3)]
        chi = [clo[d] + crse_old.shape[3 - d] - 1 for d in range(3)]
        L = self.lib
        L.orc_interp_from_coarse.argtypes = [C.c_void_p, _I3, _I3, C.c_void_p, C.c_void_p, _I3, _I3, C.c_int, _I3, _I3, C.c_double, C.c_double, C.c_int, C.c_int,
                                             C.c_int, C.c_int, _I3]
        L.orc_interp_from_coarse(fine.ctypes.data_as(C.c_void_p), _i3(flo), _i3(fhi), crse_old.ctypes.data_as(C.c_void_p), crse_new.ctypes.data_as(C.c_void_p),
                                 _i3(clo), _i3(chi), fine.shape[0], _i3(region[0]), _i3(region[1]), w_old, w_new, ncomp, method, int(hooks), ndim, _i3(ratio))

    def sim(self, problem, ndim, n_cell, prob_lo, prob_hi, periodic, max_grid_size=None, cfl=-1.0, stop_time=-1.0,
            max_timesteps=-1, reconstruction_order=-1, nscalars=0, table=None, rad_pow_mode=0, hydro1d=None, beta_order=0, c_hat_factor=0.0,
            opacity_model=0, dust_coeff=0.0) -> "OracleSim":
        n_cell = list(n_cell) + [1] * (3 - len(n_cell))
        mgs = list(max_grid_size) if max_grid_size is not None else list(n_cell)
        mgs = mgs + [1] * (3 - len(mgs))
        cfg = SimConfig(problem, ndim, _i3(n_cell), _i3(mgs), (C.c_double * 3)(*prob_lo), (C.c_double * 3)(*prob_hi), _i3(periodic),
                        cfl, stop_time, max_timesteps, reconstruction_order, nscalars)
        if table is not None:  # (r_over_r0, Erad, Frad) columns
            cols = [np.ascontiguousarray(c, dtype=np.float64) for c in table]
            cfg.table_len = len(cols[0])
            cfg.table_r, cfg.table_Erad, cfg.table_Frad = (_dp(c) for c in cols[:3])
            if len(cols) > 3:  # RADTUBE: (x, rho, Pgas, Erad)
                cfg.table_extra = _dp(cols[3])
            self._keepalive = cols
        cfg.rad_pow_mode = rad_pow_mode
        cfg.opacity_model = int(opacity_model)
        if hydro1d is not None:  # dict: gamma, profile, x_split, left, right, dirichlet, cfl, max_dt, init_dt, stop_time (see oracle/problems.hpp Hydro1DSpec)
            h = hydro1d
            vals = [h["gamma"], h.get("x_split", 0.0)] + list(h.get("left", [0, 0, 0])) + list(h.get("right", [0, 0, 0])) + \
                   [h["cfl"], h.get("max_dt", -1.0), h.get("init_dt", -1.0), h["stop_time"]]
            cfg.h1d = (C.c_double * 12)(*[float(v) for v in vals])
            cfg.h1d_i = (C.c_int * 2)(int(h.get("profile", 0)), int(h.get("dirichlet", 1)))
        if c_hat_factor > 0:  # COUPLING: the reduced-speed-of-light variant (RadMatterCouplingRSLA)
            cfg.h1d[0] = float(c_hat_factor)
        if dust_coeff > 0:  # LINE_COOLING*, MARSHAK_DUST_PE: radiation.dust_gas_interaction_coeff of the deck
            cfg.h1d[1] = float(dust_coeff)
        if beta_order > 0:  # RADSHOCK / ADVECTING parity variants: override RadSystem_Traits::beta_order
            cfg.h1d_i[0] = int(beta_order)
        # team size by problem size: ~16k cells per thread at least (a 1-D 512-cell run makes ~10^5 tiny parallel regions per second)
        ncells = int(np.prod(n_cell[:ndim]))
        self.lib.orc_set_num_threads(int(max(1, min(usable_cores(), ncells // 16384))))
        h = self.lib.orc_sim_create(C.byref(cfg))
        assert h, "oracle: unknown problem"
        return OracleSim(self, h, ndim)


@dataclass
class OracleSim:
    o: Oracle
    h: int
    ndim: int

    def __del__(self):
        try:
            self.o.lib.orc_sim_destroy(self.h)
        except Exception:
            pass

    @property
    def nboxes(self):
        return self.o.lib.orc_sim_nboxes(self.h)

    @property
    def ncomp(self):
        return self.o.lib.orc_sim_ncomp(self.h)

    @property
    def nghost(self):
        return self.o.lib.orc_sim_nghost(self.h)

    @property
    def time(self):
        return self.o.lib.orc_sim_time(self.h)

    @property
    def dt(self):
        return self.o.lib.orc_sim_dt(self.h)

    @property
    def istep(self):
        return self.o.lib.orc_sim_istep(self.h)

    def box(self, b):
        lo, hi = _I3(), _I3()
        self.o.lib.orc_sim_box(self.h, b, lo, hi)
        return list(lo), list(hi)

    def fab_shape(self, b):
        lo, hi = self.box(b)
        ng = self.nghost
        n = [hi[d] - lo[d] + 1 + (2 * ng if d < self.ndim else 0) for d in range(3)]
        return (self.ncomp, n[2], n[1], n[0])

    def state(self, b=0, which=0) -> np.ndarray:
        """Whole fab (valid + ghosts) as (ncomp, nz, ny, nx)."""
        a = np.empty(self.fab_shape(b), dtype=np.float64)
        self.o.lib.orc_sim_get_state(self.h, which, b, _dp(a))
        return a

    def valid(self, b=0, which=0) -> np.ndarray:
        a = self.state(b, which)
        ng = self.nghost
        sl = [slice(None)] + [slice(ng, -ng) if d < self.ndim else slice(None) for d in (2, 1, 0)]
        return np.ascontiguousarray(a[tuple(sl)])

    def set_state(self, a: np.ndarray, b=0, which=0):
        assert a.shape == self.fab_shape(b)
        self.o.lib.orc_sim_set_state(self.h, which, b, _dp(np.ascontiguousarray(a)))

    def fill_ghosts(self, which=0, time=0.0):
        self.o.lib.orc_sim_fill_ghosts(self.h, which, time)

    def compute_dt(self) -> float:
        return self.o.lib.orc_sim_compute_dt(self.h)

    def step(self) -> bool:
        return bool(self.o.lib.orc_sim_step(self.h))

    def advance_fixed_dt(self, dt: float) -> bool:
        return bool(self.o.lib.orc_sim_advance_fixed_dt(self.h, dt))

    def evolve(self) -> bool:
        return bool(self.o.lib.orc_sim_evolve(self.h))

    def set_limits(self, density_floor: float = 0.0, temp_floor: float = 0.0):
        """the floors EnforceLimits applies after every stage (0: none, the default)"""
        self.o.lib.orc_sim_set_limits.argtypes = [C.c_void_p, C.c_double, C.c_double]
        self.o.lib.orc_sim_set_limits(self.h, float(density_floor), float(temp_floor))

    def set_fused_fluxes(self, on: bool, stages: bool = True):
        """the fused, vectorised flux evaluation (oracle/hydro_fused.hpp) instead of the operator sequence: the same bits, one pass per box;
        stages: the update, the limits and the dual-energy sync of a stage in the same pass over the box as well (HydroSim::fusedStage)"""
        self.o.lib.orc_sim_set_fused_fluxes.argtypes = [C.c_void_p, C.c_int]
        self.o.lib.orc_sim_set_fused_fluxes(self.h, (1 if on else 0) | (2 if (on and stages) else 0))

    def hydro_fluxes(self, b: int, direction: int, fused: bool):
        """(flux[6, faces...], face velocity) of box b from the current state_new (ghost cells filled here), by either form"""
        lo, hi = self.box(b)
        n = [hi[d] - lo[d] + 1 + (1 if d == direction else 0) for d in range(3)]
        F = np.empty((6, n[2], n[1], n[0]))
        V = np.empty((n[2], n[1], n[0]))
        f = self.o.lib.orc_sim_hydro_fluxes
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        assert f(self.h, int(bool(fused)), int(b), int(direction), _dp(F), _dp(V)) == V.size
        return F, V

    def rad_transport_only(self, dt_radiation: float):
        self.o.lib.orc_sim_rad_transport_only(self.h, float(dt_radiation))

    def set_rad_reconstruction_order(self, order: int):
        self.o.lib.orc_sim_set_rad_reconstruction_order.argtypes = [C.c_void_p, C.c_int]
        self.o.lib.orc_sim_set_rad_reconstruction_order(self.h, int(order))

    def set_wavespeed_correction(self, on: bool):
        """QuokkaSimulation::use_wavespeed_correction_ (reference src/QuokkaSimulation.hpp:133): ComputeCellOpticalDepth + S_corr on even faces"""
        self.o.lib.orc_sim_set_wavespeed_correction.argtypes = [C.c_void_p, C.c_int]
        self.o.lib.orc_sim_set_wavespeed_correction(self.h, int(bool(on)))

    def run_record(self, nsteps: int, b=0, cell=(0, 0, 0)):
        """nsteps steps; returns (times, states[nsteps, ncomp]) of one valid cell after every step"""
        t = np.zeros(nsteps)
        u = np.zeros((nsteps, self.ncomp))
        f = self.o.lib.orc_sim_run_record
        f.restype = C.c_long
        f.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        n = f(self.h, nsteps, int(b), int(cell[0]), int(cell[1]), int(cell[2]), _dp(t), _dp(u))
        return t[:n], u[:n]

    def rad_counters(self):
        out = (C.c_long * 8)()
        self.o.lib.orc_sim_rad_counters(self.h, out)
        keys = ["solves", "newton_iterations", "max_newton_iterations", "decoupled", "fail_coupling", "fail_dust", "fail_outer", "rad_cell_updates"]
        return dict(zip(keys, list(out)))

    def tag_relative_gradient(self, b: int, field: int, eta_threshold: float, q_min: float, min_inclusive: bool) -> np.ndarray:
        """ErrorEst (gradient-threshold family) on the ghost-filled new state of box b: int8 array over the valid box (2 = TagBox::SET)"""
        lo, hi = self.box(b)
        t = np.zeros((hi[2] - lo[2] + 1, hi[1] - lo[1] + 1, hi[0] - lo[0] + 1), dtype=np.int8)
        self.o.lib.orc_sim_tag_relative_gradient(self.h, b, int(field), C.c_double(eta_threshold), C.c_double(q_min), int(bool(min_inclusive)),
                                                 t.ctypes.data_as(C.c_void_p))
        return t

    def tag_centered_gradient(self, b: int, comp: int, direction: int, dx: float, eta_threshold: float, q_min: float, min_inclusive: bool) -> np.ndarray:
        """ErrorEst of HydroShocktube (centred difference / (2 dx)) on the ghost-filled new state of box b"""
        lo, hi = self.box(b)
        t = np.zeros((hi[2] - lo[2] + 1, hi[1] - lo[1] + 1, hi[0] - lo[0] + 1), dtype=np.int8)
        self.o.lib.orc_sim_tag_centered_gradient(self.h, b, int(comp), int(direction), C.c_double(dx), C.c_double(eta_threshold), C.c_double(q_min),
                                                 int(bool(min_inclusive)), t.ctypes.data_as(C.c_void_p))
        return t

    def rad_source(self, b=0, time=0.0) -> np.ndarray:
        lo, hi = self.box(b)
        a = np.empty((hi[2] - lo[2] + 1, hi[1] - lo[1] + 1, hi[0] - lo[0] + 1), dtype=np.float64)
        self.o.lib.orc_sim_rad_source(self.h, b, C.c_double(time), _dp(a))
        return a

    def counters(self):
        out = (C.c_long * 3)()
        self.o.lib.orc_sim_counters(self.h, out)
        return {"fofc1_cells": out[0], "fofc2_cells": out[1], "retries": out[2]}


class OracleCloudy:
    """oracle/cooling.hpp: the reference's tabulated cooling on the arrays of a cloudy_cooling_tools file (Parameter1, Temperature, Cooling,
    Heating, MMW as H5Dread delivers them)"""
    TGAS_FROM_EGAS, EGAS_FROM_TGAS, MMW, COOLING_LENGTH, NET_HEATING = range(5)

    def __init__(self, arrays: dict, variant: str = "direct"):
        self.o = Oracle(variant)
        L = self.lib = self.o.lib
        L.orc_cloudy_create.restype = C.c_void_p
        DP = C.POINTER(C.c_double)
        L.orc_cloudy_create.argtypes = [C.c_int, C.c_int, DP, DP, DP, DP, DP]
        L.orc_cloudy_destroy.argtypes = [C.c_void_p]
        L.orc_cloudy_ranges.argtypes = [C.c_void_p, DP]
        L.orc_cloudy_get.argtypes = [C.c_void_p, C.c_int, DP]
        L.orc_cloudy_evaluate.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_long, DP, DP, DP]
        L.orc_cloudy_compute_cooling.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_long, DP, C.POINTER(C.c_int)]
        a = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in arrays.items()}
        self.n0, self.n1 = a["Cooling"].shape
        self.h = C.c_void_p(L.orc_cloudy_create(self.n0, self.n1, _dp(a["Parameter1"]), _dp(a["Temperature"]), _dp(a["Cooling"]), _dp(a["Heating"]), _dp(a["MMW"])))

    def __del__(self):
        try:
            self.lib.orc_cloudy_destroy(self.h)
        except Exception:
            pass

    def ranges(self):
        """(T_min, T_max, mmw_min, mmw_max)"""
        r = np.zeros(4)
        self.lib.orc_cloudy_ranges(self.h, _dp(r))
        return tuple(float(v) for v in r)

    def prepared(self, which: int) -> np.ndarray:
        """0 log_nH, 1 log_Tgas, 2 cooling, 3 heating, 4 mean molecular weight (n_H fastest)"""
        out = np.zeros([self.n0, self.n1, self.n0 * self.n1, self.n0 * self.n1, self.n0 * self.n1][which])
        self.lib.orc_cloudy_get(self.h, which, _dp(out))
        return out

    def evaluate(self, what: int, rho, val, gamma: float) -> np.ndarray:
        rho = np.ascontiguousarray(rho, dtype=np.float64)
        val = np.ascontiguousarray(val, dtype=np.float64)
        out = np.zeros_like(rho)
        self.lib.orc_cloudy_evaluate(self.h, gamma, what, rho.size, _dp(rho), _dp(val), _dp(out))
        return out

    def compute_cooling(self, U: np.ndarray, gamma: float, dt: float, T_floor: float):
        """U[6, n] = (rho, x1Mom, x2Mom, x3Mom, Egas, Eint_aux): returns (U after computeCooling, substeps per cell)"""
        U = np.ascontiguousarray(U, dtype=np.float64).copy()
        n = U.shape[1]
        ns = np.zeros(n, dtype=np.int32)
        self.lib.orc_cloudy_compute_cooling(self.h, gamma, dt, T_floor, n, _dp(U), ns.ctypes.data_as(C.POINTER(C.c_int)))
        return U, ns
