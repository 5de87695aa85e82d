"""Thin Python view of the reference's operator surface over the C-ABI (same names, same argument
meaning as HyperbolicSystem<problem_t> / HydroSystem<problem_t>; reference file:line per entry in
include/quokka_amd.h).  Used by the test-suite and bench harness; the C++ host mirror lives in
quokka_amd/host/.  No arithmetic happens here.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch

from . import capi
from .multifab import Level, MultiFab


def _p3(mfs: Sequence[MultiFab]):
    arr = (C.c_void_p * 3)()
    for d in range(3):
        arr[d] = mfs[d].ptr if d < len(mfs) and mfs[d] is not None else None
    return arr


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in (list(v) + [1.0, 1.0, 1.0])[:3]])


class HyperbolicSystem:
    @staticmethod
    def ReconstructStatesConstant(lev: Level, DIR: int, q: MultiFab, leftState: MultiFab, rightState: MultiFab, nghost: int, nvars: int):
        c = lev.ctx
        c.check(c.L.qk_ReconstructStatesConstant(lev.h, c.stream(), DIR, q.ptr, leftState.ptr, rightState.ptr, nghost, nvars), "ReconstructStatesConstant")

    @staticmethod
    def ReconstructStatesPLM(lev: Level, DIR: int, limiter: int, q: MultiFab, leftState: MultiFab, rightState: MultiFab, nghost: int, nvars: int):
        c = lev.ctx
        c.check(c.L.qk_ReconstructStatesPLM(lev.h, c.stream(), DIR, limiter, q.ptr, leftState.ptr, rightState.ptr, nghost, nvars), "ReconstructStatesPLM")

    @staticmethod
    def ReconstructStatesPPM(lev: Level, DIR: int, q: MultiFab, leftState: MultiFab, rightState: MultiFab, nghost: int, nvars: int,
                             iReadFrom: int = 0, iWriteFrom: int = 0):
        c = lev.ctx
        c.check(c.L.qk_ReconstructStatesPPM(lev.h, c.stream(), DIR, q.ptr, leftState.ptr, rightState.ptr, nghost, nvars, iReadFrom, iWriteFrom),
                "ReconstructStatesPPM")


class HydroSystem(HyperbolicSystem):
    """HydroSystem<problem_t>: `traits` carries what the problem's trait structs carry."""

    def __init__(self, traits: capi.HydroTraits):
        self.traits = traits
        self.nvar_ = 6 + traits.nscalars

    def _t(self):
        return C.byref(self.traits)

    def ConservedToPrimitive(self, lev, cons, primVar, nghost):
        c = lev.ctx
        c.check(c.L.qk_hydro_ConservedToPrimitive(lev.h, c.stream(), self._t(), cons.ptr, primVar.ptr, nghost), "ConservedToPrimitive")

    def ComputeFlatteningCoefficients(self, lev, DIR, primVar, x1Chi, nghost):
        c = lev.ctx
        c.check(c.L.qk_hydro_ComputeFlatteningCoefficients(lev.h, c.stream(), self._t(), DIR, primVar.ptr, x1Chi.ptr, nghost),
                "ComputeFlatteningCoefficients")

    def FlattenShocks(self, lev, DIR, q, x1Chi, x2Chi, x3Chi, x1LeftState, x1RightState, nghost, nvars):
        c = lev.ctx
        c.check(c.L.qk_hydro_FlattenShocks(lev.h, c.stream(), self._t(), DIR, q.ptr, x1Chi.ptr, x2Chi.ptr if x2Chi else None,
                                           x3Chi.ptr if x3Chi else None, x1LeftState.ptr, x1RightState.ptr, nghost, nvars), "FlattenShocks")

    def ComputeFluxes(self, lev, RIEMANN, DIR, x1Flux, x1FaceVel, x1LeftState, x1RightState, primVar, K_visc):
        c = lev.ctx
        c.check(c.L.qk_hydro_ComputeFluxes(lev.h, c.stream(), self._t(), RIEMANN, DIR, x1Flux.ptr, x1FaceVel.ptr, x1LeftState.ptr,
                                           x1RightState.ptr, primVar.ptr, float(K_visc)), "ComputeFluxes")

    def ComputeRhsFromFluxes(self, lev, rhs, fluxArray, dx, nvars):
        c = lev.ctx
        c.check(c.L.qk_hydro_ComputeRhsFromFluxes(lev.h, c.stream(), self._t(), rhs.ptr, _p3(fluxArray), _d3(dx), nvars), "ComputeRhsFromFluxes")

    def AddInternalEnergyPdV(self, lev, rhs, consVar, dx, faceVelArray, redoFlag):
        c = lev.ctx
        c.check(c.L.qk_hydro_AddInternalEnergyPdV(lev.h, c.stream(), self._t(), rhs.ptr, consVar.ptr, _d3(dx), _p3(faceVelArray), redoFlag.ptr),
                "AddInternalEnergyPdV")

    def PredictStep(self, lev, consVarOld, consVarNew, rhs, dt, nvars, redoFlag, redo_count: torch.Tensor = None):
        c = lev.ctx
        cnt = C.c_void_p(redo_count.data_ptr()) if redo_count is not None else None
        c.check(c.L.qk_hydro_PredictStep(lev.h, c.stream(), self._t(), consVarOld.ptr, consVarNew.ptr, rhs.ptr, float(dt), nvars, redoFlag.ptr, cnt),
                "PredictStep")

    def EnforceLimits(self, lev, densityFloor, tempFloor, state):
        c = lev.ctx
        c.check(c.L.qk_hydro_EnforceLimits(lev.h, c.stream(), self._t(), float(densityFloor), float(tempFloor), state.ptr), "EnforceLimits")

    def SyncDualEnergy(self, lev, consVar, error_flag: torch.Tensor = None):
        c = lev.ctx
        ef = C.c_void_p(error_flag.data_ptr()) if error_flag is not None else None
        c.check(c.L.qk_hydro_SyncDualEnergy(lev.h, c.stream(), self._t(), consVar.ptr, ef), "SyncDualEnergy")

    def ComputeMaxSignalSpeed(self, lev, cons, maxSignal):
        c = lev.ctx
        c.check(c.L.qk_hydro_ComputeMaxSignalSpeed(lev.h, c.stream(), self._t(), cons.ptr, maxSignal.ptr), "ComputeMaxSignalSpeed")

    def maxSignalSpeedLocal(self, lev, cons, which: int = 0, out: torch.Tensor = None) -> torch.Tensor:
        c = lev.ctx
        if out is None:
            out = torch.zeros(1, dtype=torch.float64, device=c.device)
        c.check(c.L.qk_hydro_maxSignalSpeedLocal(lev.h, c.stream(), self._t(), which, cons.ptr, C.c_void_p(out.data_ptr())), "maxSignalSpeedLocal")
        return out


def replaceFluxes(lev, DIR, flux, FOflux, redoFlag, face_ncomp):
    c = lev.ctx
    c.check(c.L.qk_replaceFluxes(lev.h, c.stream(), DIR, flux.ptr, FOflux.ptr, redoFlag.ptr, face_ncomp), "replaceFluxes")


def Saxpy(lev, DIR, dst, a, src, ncomp):
    c = lev.ctx
    c.check(c.L.qk_Saxpy(lev.h, c.stream(), DIR, dst.ptr, float(a), src.ptr, ncomp), "Saxpy")
