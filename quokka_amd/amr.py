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


def TagBoxArray(lev: Level) -> MultiFab:
    """amrex::TagBoxArray: one char per cell, no ghost cells, cleared"""
    return MultiFab(lev, 1, 0, dtype=torch.int8, fill=capi.TAG_CLEAR)


def tag_relative_gradient(lev: Level, traits: capi.HydroTraits, state: MultiFab, tags: MultiFab, field: int, eta_threshold: float, q_min: float,
                          min_inclusive: bool):
    ctx = lev.ctx
    ctx.check(ctx.L.qk_tag_relative_gradient(lev.h, ctx.stream(), C.byref(traits), state.ptr, tags.ptr, int(field), float(eta_threshold), float(q_min),
                                             int(bool(min_inclusive))), "qk_tag_relative_gradient")


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
