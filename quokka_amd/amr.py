"""AMR level machinery over the C-ABI — the data-parallel pieces built so far:

  tag_relative_gradient   QuokkaSimulation<problem_t>::ErrorEst of the gradient-threshold family
                          (reference src/problems/HydroBlast3D/test_hydro3d_blast.cpp:118-151, RadhydroShell:337-371)
  Pre/PostInterpState     QuokkaSimulation::PreInterpState / PostInterpState (reference src/QuokkaSimulation.hpp:804-841)
  AverageDown             AMRSimulation::AverageDownTo -> amrex::average_down (reference src/simulation.hpp:1949-1964)

Grid generation, FillPatch interpolation, flux registers and the subcycling driver are not built yet.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import capi
from .multifab import Level, MultiFab


def _box_array(boxes):
    """(n, ctypes array | None) for an optional global box list"""
    if boxes is None:
        return 0, None
    arr = (capi.Box * max(len(boxes), 1))(*[capi.Box((C.c_int * 3)(*[int(x) for x in lo]), (C.c_int * 3)(*[int(x) for x in hi])) for lo, hi in boxes])
    return len(boxes), arr


def TagBoxArray(lev: Level) -> MultiFab:
    """amrex::TagBoxArray: one char per cell, no ghost cells, cleared"""
    return MultiFab(lev, 1, 0, dtype=torch.int8, fill=capi.TAG_CLEAR)


def tag_relative_gradient(lev: Level, traits: capi.HydroTraits, state: MultiFab, tags: MultiFab, field: int, eta_threshold: float, q_min: float,
                          min_inclusive: bool):
    ctx = lev.ctx
    ctx.check(ctx.L.qk_tag_relative_gradient(lev.h, ctx.stream(), C.byref(traits), state.ptr, tags.ptr, int(field), float(eta_threshold), float(q_min),
                                             int(bool(min_inclusive))), "qk_tag_relative_gradient")


def tag_centered_gradient(lev: Level, state: MultiFab, tags: MultiFab, comp: int, direction: int, dx: float, eta_threshold: float, q_min: float,
                          min_inclusive: bool):
    """ErrorEst of HydroShocktube (reference src/problems/HydroShocktube/test_hydro_shocktube.cpp:146-170)"""
    ctx = lev.ctx
    ctx.check(ctx.L.qk_tag_centered_gradient(lev.h, ctx.stream(), state.ptr, tags.ptr, int(comp), int(direction), float(dx), float(eta_threshold), float(q_min),
                                             int(bool(min_inclusive))), "qk_tag_centered_gradient")


def PreInterpState(lev: Level, mf: MultiFab):
    lev.ctx.check(lev.ctx.L.qk_PreInterpState(lev.h, lev.ctx.stream(), mf.ptr), "qk_PreInterpState")


def PostInterpState(lev: Level, mf: MultiFab):
    lev.ctx.check(lev.ctx.L.qk_PostInterpState(lev.h, lev.ctx.stream(), mf.ptr), "qk_PostInterpState")


class AverageDown:
    def __init__(self, crse: Level, fine: Level, ratio=(2, 2, 2)):
        self.crse, self.fine = crse, fine
        h = C.c_void_p()
        crse.ctx.check(crse.ctx.L.qk_avgdown_plan_create(crse.h, fine.h, (C.c_int * 3)(*ratio), C.byref(h)), "qk_avgdown_plan_create")
        self.h = h

    def num_items(self) -> int:
        return self.crse.ctx.L.qk_avgdown_plan_num_items(self.h)

    def __call__(self, fine_mf: MultiFab, crse_mf: MultiFab, scomp: int, ncomp: int):
        ctx = self.crse.ctx
        ctx.check(ctx.L.qk_average_down(self.h, ctx.stream(), fine_mf.ptr, crse_mf.ptr, scomp, ncomp), "qk_average_down")

    def __del__(self):
        try:
            self.crse.ctx.L.qk_avgdown_plan_destroy(self.h)
        except Exception:
            pass


class InterpFromCoarse:
    """coarse -> fine part of FillPatchTwoLevels (reference src/simulation.hpp:1789-1858): interpolates the fine cells that no fine
    box covers (whole_fab: every cell of the grown fine boxes) from w_old * crse_old + w_new * crse_new"""

    def __init__(self, crse: Level, fine: Level, fine_geom, nghost: int, ratio=(2, 2, 2), whole_fab: bool = False, all_fine_boxes=None):
        self.crse, self.fine = crse, fine
        self._geom_c = fine_geom.c_struct()
        h = C.c_void_p()
        n_all, arr = _box_array(all_fine_boxes)
        crse.ctx.check(crse.ctx.L.qk_interp_plan_create(crse.h, fine.h, C.byref(self._geom_c), nghost, (C.c_int * 3)(*ratio), int(whole_fab), n_all, arr,
                                                        C.byref(h)), "qk_interp_plan_create")
        self.h = h

    def items(self):
        L, out = self.crse.ctx.L, []
        for idx in range(L.qk_interp_plan_num_items(self.h)):
            fb, cb = C.c_int(), C.c_int()
            lo, hi = (C.c_int * 3)(), (C.c_int * 3)()
            self.crse.ctx.check(L.qk_interp_plan_item(self.h, idx, C.byref(fb), C.byref(cb), lo, hi), "qk_interp_plan_item")
            out.append((fb.value, cb.value, list(lo), list(hi)))
        return out

    def __call__(self, fine_mf: MultiFab, crse_old: MultiFab, crse_new: MultiFab, w_old: float, w_new: float, ncomp: int, method: int = 1,
                 energy_hooks: bool = True):
        ctx = self.crse.ctx
        ctx.check(ctx.L.qk_InterpFromCoarse(self.h, ctx.stream(), fine_mf.ptr, crse_old.ptr, crse_new.ptr, float(w_old), float(w_new), ncomp, method,
                                            int(energy_hooks)), "qk_InterpFromCoarse")

    def __del__(self):
        try:
            self.crse.ctx.L.qk_interp_plan_destroy(self.h)
        except Exception:
            pass


class FluxRegister:
    """amrex::YAFluxRegister between `crse` and the next finer level `fine` (reference src/simulation.hpp:1345-1387, :1308)"""

    def __init__(self, crse: Level, fine: Level, crse_geom, ncomp: int, ratio=(2, 2, 2), all_fine_boxes=None, reg_nghost: int = 0):
        self.crse, self.fine, self.ncomp = crse, fine, ncomp
        self._geom_c = crse_geom.c_struct()
        h = C.c_void_p()
        n_all, arr = _box_array(all_fine_boxes)
        crse.ctx.check(crse.ctx.L.qk_fluxreg_create(crse.h, fine.h, C.byref(self._geom_c), (C.c_int * 3)(*ratio), ncomp, n_all, arr, reg_nghost, C.byref(h)),
                       "qk_fluxreg_create")
        self.h = h

    def items(self):
        L, out = self.crse.ctx.L, []
        for idx in range(L.qk_fluxreg_num_items(self.h)):
            d, sd, fb, cb = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            lo, hi, sh = (C.c_int * 3)(), (C.c_int * 3)(), (C.c_int * 3)()
            self.crse.ctx.check(L.qk_fluxreg_item(self.h, idx, C.byref(d), C.byref(sd), C.byref(fb), C.byref(cb), lo, hi, sh), "qk_fluxreg_item")
            out.append((d.value, sd.value, fb.value, cb.value, list(lo), list(hi), list(sh)))
        return out

    @staticmethod
    def _p3(mfs):
        arr = (C.c_void_p * 3)()
        for d in range(3):
            arr[d] = mfs[d].ptr if d < len(mfs) else None
        return arr

    def reset(self):
        ctx = self.crse.ctx
        ctx.check(ctx.L.qk_fluxreg_reset(self.h, ctx.stream()), "qk_fluxreg_reset")

    def save(self):
        """the register as it is before a level's retry loop (reference src/QuokkaSimulation.hpp:894-900)"""
        ctx = self.crse.ctx
        ctx.check(ctx.L.qk_fluxreg_save(self.h, ctx.stream()), "qk_fluxreg_save")

    def restore(self):
        ctx = self.crse.ctx
        ctx.check(ctx.L.qk_fluxreg_restore(self.h, ctx.stream()), "qk_fluxreg_restore")

    def CrseAdd(self, flux, dx, dt: float):
        ctx = self.crse.ctx
        ctx.check(ctx.L.qk_fluxreg_CrseAdd(self.h, ctx.stream(), self._p3(flux), (C.c_double * 3)(*[float(x) for x in dx]), float(dt)), "qk_fluxreg_CrseAdd")

    def FineAdd(self, flux, dx_fine, dt: float):
        ctx = self.crse.ctx
        ctx.check(ctx.L.qk_fluxreg_FineAdd(self.h, ctx.stream(), self._p3(flux), (C.c_double * 3)(*[float(x) for x in dx_fine]), float(dt)), "qk_fluxreg_FineAdd")

    def set_state_component(self, comp0: int):
        """register component n <-> state component comp0 + n in Reflux (the radiation block of a radiation-hydro state)"""
        ctx = self.crse.ctx
        ctx.check(ctx.L.qk_fluxreg_set_state_component(self.h, int(comp0)), "qk_fluxreg_set_state_component")

    def Reflux(self, crse_state: MultiFab):
        ctx = self.crse.ctx
        ctx.check(ctx.L.qk_fluxreg_Reflux(self.h, ctx.stream(), crse_state.ptr), "qk_fluxreg_Reflux")

    def __del__(self):
        try:
            self.crse.ctx.L.qk_fluxreg_destroy(self.h)
        except Exception:
            pass
