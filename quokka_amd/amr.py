"""AMR level machinery over the C-ABI — the data-parallel pieces built so far:

  tag_relative_gradient   QuokkaSimulation<problem_t>::ErrorEst of the gradient-threshold family
                          (reference src/problems/HydroBlast3D/test_hydro3d_blast.cpp:118-151, RadhydroShell:337-371)
  Pre/PostInterpState     QuokkaSimulation::PreInterpState / PostInterpState (reference src/QuokkaSimulation.hpp:804-841)
  AverageDown             AMRSimulation::AverageDownTo -> amrex::average_down (reference src/simulation.hpp:1949-1964)

  InterpFromCoarse        coarse -> fine part of FillPatchTwoLevels (:1789-1858);  FluxRegister  amrex::YAFluxRegister (:1345-1387, :1308)
  ParallelCopy            amrex::FabArray::ParallelCopy / ParallelAdd between two box layouts with independent owners
  DistFluxRegister        YAFluxRegister with the coarse part on the coarse level's ranks and the fine part on the fine level's

The subcycling driver, grid generation and the distribution of the levels over the ranks live in amr_simulation.py.
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


class ParallelCopy:
    """amrex::FabArray::ParallelCopy (add=False) / ParallelAdd (add=True) between two box layouts (qk_pcopy_plan, csrc/qk_amr_pcopy.hip): the
    box lists and owners describe ALL ranks; the MultiFabs passed to a call hold this rank's boxes of each list, in list order.  Same wire
    protocol as GhostExchange: pack -> one send / recv pair per peer (RCCL point-to-point) -> same-rank copies -> wait -> unpack."""

    def __init__(self, ctx, geom, src_boxes, src_owner, dst_boxes, dst_owner, ncomp: int, rank: int, src_nghost: int = 0, dst_nghost: int = 0,
                 src_ring_only: bool = False, dst_holes=None):
        self.ctx, self.ncomp = ctx, ncomp
        self._geom_c = geom.c_struct()
        ns, sarr = _box_array(src_boxes)
        nd, darr = _box_array(dst_boxes)
        so = (C.c_int * max(ns, 1))(*[int(o) for o in src_owner])
        do = (C.c_int * max(nd, 1))(*[int(o) for o in dst_owner])
        holes = None
        if dst_holes is not None:
            _, holes = _box_array(dst_holes)
        h = C.c_void_p()
        ctx.check(ctx.L.qk_pcopy_plan_create(ctx.h, C.byref(self._geom_c), ns, sarr, so, src_nghost, int(src_ring_only), nd, darr, do, dst_nghost,
                                             None if holes is None else C.cast(holes, C.c_void_p), ncomp, rank, C.byref(h)), "qk_pcopy_plan_create")
        self.h = h
        self.peers = []
        for k in range(ctx.L.qk_pcopy_plan_num_peers(h)):
            r, nsend, nrecv = C.c_int(), C.c_int64(), C.c_int64()
            ctx.check(ctx.L.qk_pcopy_plan_peer(h, k, C.byref(r), C.byref(nsend), C.byref(nrecv)), "qk_pcopy_plan_peer")
            self.peers.append((k, r.value, torch.empty(nsend.value, dtype=torch.float64, device=ctx.device),
                               torch.empty(nrecv.value, dtype=torch.float64, device=ctx.device)))

    def items(self, kind: int, k: int = 0):
        """plan introspection: (dst_box, src_box, lo, hi, shift, offset); kind 0 same-rank, 1 packed for peer k, 2 unpacked from peer k"""
        L, out = self.ctx.L, []
        for idx in range(L.qk_pcopy_plan_num_items(self.h, kind, k)):
            db, sb, off = C.c_int(), C.c_int(), C.c_int64()
            lo, hi, sh = (C.c_int * 3)(), (C.c_int * 3)(), (C.c_int * 3)()
            self.ctx.check(L.qk_pcopy_plan_item(self.h, kind, k, idx, C.byref(db), C.byref(sb), lo, hi, sh, C.byref(off)), "qk_pcopy_plan_item")
            out.append((db.value, sb.value, list(lo), list(hi), list(sh), off.value))
        return out

    def __call__(self, src: MultiFab, dst: MultiFab, scomp_src: int = 0, scomp_dst: int = 0, add: bool = False):
        ctx = self.ctx
        L, s = ctx.L, ctx.stream()
        self.run(lambda k, sbuf: ctx.check(L.qk_ParallelCopy_pack(self.h, s, k, src.ptr, scomp_src, C.c_void_p(sbuf.data_ptr())), "ParallelCopy_pack"),
                 lambda: ctx.check(L.qk_ParallelCopy_local(self.h, s, src.ptr, dst.ptr, scomp_src, scomp_dst, int(add)), "ParallelCopy_local"),
                 lambda k, rbuf: ctx.check(L.qk_ParallelCopy_unpack(self.h, s, k, dst.ptr, scomp_dst, C.c_void_p(rbuf.data_ptr()), int(add)), "ParallelCopy_unpack"))

    def run(self, pack, local, unpack):
        """the protocol, independent of who moves the bytes inside a rank (HIP kernels in the product, numpy in the gloo tests on CPU)"""
        pending = None
        if self.peers:
            from . import comm
            for k, r, sbuf, rbuf in self.peers:
                if sbuf.numel():
                    pack(k, sbuf)
            pending = comm.exchange([(r, sbuf, rbuf) for k, r, sbuf, rbuf in self.peers])
        local()
        if pending is not None:
            pending.wait()
        for k, r, sbuf, rbuf in self.peers:
            if rbuf.numel():
                unpack(k, rbuf)

    def __del__(self):
        try:
            self.ctx.L.qk_pcopy_plan_destroy(self.h)
        except Exception:
            pass


class FluxRegister:
    """amrex::YAFluxRegister between `crse` and the next finer level `fine` (reference src/simulation.hpp:1345-1387, :1308)"""

    def __init__(self, crse: Level, fine: Level, crse_geom, ncomp: int, ratio=(2, 2, 2), all_fine_boxes=None, reg_nghost: int = 0, crse_part: bool = False):
        self.crse, self.fine, self.ncomp = crse, fine, ncomp
        self._geom_c = crse_geom.c_struct()
        h = C.c_void_p()
        if crse_part:  # `fine` holds the fine boxes of ALL ranks; register cells outside the local coarse boxes belong to their owners
            crse.ctx.check(crse.ctx.L.qk_fluxreg_create_crse_part(crse.h, fine.h, C.byref(self._geom_c), (C.c_int * 3)(*ratio), ncomp, C.byref(h)),
                           "qk_fluxreg_create_crse_part")
        else:
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


class DistFluxRegister:
    """amrex::YAFluxRegister when the fine level has a DistributionMapping of its own (reference src/simulation.hpp:1421-1500): the coarse part
    (m_crse_data) lives with the coarse boxes and takes CrseAdd; the fine part (m_cfpatch) lives with the fine boxes — register cells in the
    one-cell ghost ring of the coarsened fine boxes (`shadow_lev`) — and takes FineAdd; Reflux adds the coarse part to the local coarse state and
    sends the ring to the coarse owners (ParallelAdd).  Same interface as FluxRegister."""

    def __init__(self, ctx, crse: Level, crse_geom, crse_boxes, crse_owner, fine: Level, fine_boxes, fine_owner, shadow_lev: Level, shadow_boxes, ncomp: int,
                 rank: int, state_comp0: int = 0):
        self.ncomp, self.comp0 = ncomp, state_comp0
        self._all_fine = Level(ctx, crse.ndim, fine_boxes)  # box metadata of the whole fine level
        self.crse_part = FluxRegister(crse, self._all_fine, crse_geom, ncomp, crse_part=True)
        self.fine_part = FluxRegister(shadow_lev, fine, crse_geom, ncomp, all_fine_boxes=fine_boxes, reg_nghost=1)
        if state_comp0:
            self.crse_part.set_state_component(state_comp0)
        self.inc = MultiFab(shadow_lev, ncomp, 1, fill=0.0)
        self.to_crse = ParallelCopy(ctx, crse_geom, shadow_boxes, fine_owner, crse_boxes, crse_owner, ncomp, rank, src_nghost=1, src_ring_only=True)

    def items(self):
        return self.crse_part.items()  # (the coarse cells of the register on this rank: what the flux mask of the carried form marks)

    def reset(self):
        self.crse_part.reset()
        self.fine_part.reset()

    def save(self):
        self.fine_part.save()

    def restore(self):
        self.fine_part.restore()

    def CrseAdd(self, flux, dx, dt: float):
        self.crse_part.CrseAdd(flux, dx, dt)

    def FineAdd(self, flux, dx_fine, dt: float):
        self.fine_part.FineAdd(flux, dx_fine, dt)

    def Reflux(self, crse_state: MultiFab):
        self.crse_part.Reflux(crse_state)
        self.inc.storage.zero_()
        self.fine_part.Reflux(self.inc)
        self.to_crse(self.inc, crse_state, 0, self.comp0, add=True)
