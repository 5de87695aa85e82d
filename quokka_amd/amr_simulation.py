"""Block-structured AMR driver over the C-ABI: the level machinery of AMRSimulation / QuokkaSimulation

  timeStepWithSubcycling            reference src/simulation.hpp:1220-1343   (recursive subcycling, refinement ratio 2)
  regrid / MakeNewLevelFromCoarse / RemakeLevel   :1656-1702 + amrex::AmrCore (ErrorEst -> tags -> grids)
  FillPatch / fillBoundaryConditions  :1704-1858                             (fine-fine copy + coarse interpolation + physical BCs)
  incrementFluxRegisters / Reflux / AverageDownTo / FixupState   :1345-1387, :1308, :1949-1964, src/QuokkaSimulation.hpp:761-770
  computeTimestep over levels        :744-818

Host orchestration only: every cell is touched by a kernel behind include/quokka_amd.h (quokka_amd/amr.py wraps them).
Grid generation: Berger-Rigoutsos clustering with amr.grid_eff on the device-buffered tags (qk_amr_cluster_berger_rigoutsos; "tiles" keeps the
round-1 rule).  Several ranks: every level has a box -> rank map of its own (space-filling curve over the level's boxes, a level chopped until every
rank owns a box: distribute_sfc / chop_grids, as AMReX hands AMRSimulation a BoxArray + DistributionMapping per level, reference
src/simulation.hpp:1421-1500, :1657-1702); coarse data reach the fine boxes, averaged and refluxed data the coarse boxes, through CoarseShadow.
With rad_traits the levels are RadAmrLevelSim: hydro advance + radiation subcycle + a second flux register for the radiation block.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import capi
from .amr import AverageDown, DistFluxRegister, FluxRegister, InterpFromCoarse, ParallelCopy
from .multifab import Context, Level, MultiFab
from .radhydro import RAD0, RadhydroSimulation
from .simulation import NGHOST_CC, Geometry, GhostExchange, HydroSimulation, chop_domain

Box = Tuple[List[int], List[int]]


# ------------------------------------------------------------------------------------------------ grid generation (host, numpy)
def dilate(mask: np.ndarray, n: int, ndim: int) -> np.ndarray:
    """tag buffering (amr.n_error_buf): every cell within n cells (max norm) of a tag; mask is indexed [k, j, i]"""
    out = mask.copy()
    for ax in range(3):
        d = 2 - ax
        if d >= ndim or n == 0:
            continue
        acc = out.copy()
        for s in range(1, n + 1):
            sl_to, sl_from = [slice(None)] * 3, [slice(None)] * 3
            sl_to[ax], sl_from[ax] = slice(s, None), slice(None, -s)
            acc[tuple(sl_to)] |= out[tuple(sl_from)]
            acc[tuple(sl_from)] |= out[tuple(sl_to)]
        out = acc
    return out


def boxes_from_tags(tags: np.ndarray, ndim: int, n_error_buf: int, blocking_factor: int, max_grid_size: int, ratio: int = 2,
                    allowed: Optional[np.ndarray] = None) -> List[Box]:
    """tags[k, j, i] (bool, over the coarse level's index space) -> fine boxes (fine index space): buffered tags, blocking-factor tiles,
    greedy merge.  `allowed` (same shape): coarse cells a fine box may cover (proper nesting); a tile is kept only if it is allowed."""
    buf = dilate(tags.astype(bool), n_error_buf, ndim)
    tile = [max(blocking_factor // ratio, 1) if d < ndim else 1 for d in range(3)]  # tile edge in coarse cells
    nz, ny, nx = buf.shape
    for d, n in enumerate((nx, ny, nz)):
        assert n % tile[d] == 0, "domain must be divisible by blocking_factor / ratio"
    tz, ty, tx = nz // tile[2], ny // tile[1], nx // tile[0]
    t = buf.reshape(tz, tile[2], ty, tile[1], tx, tile[0]).any(axis=(1, 3, 5))
    if allowed is not None:
        t &= allowed.reshape(tz, tile[2], ty, tile[1], tx, tile[0]).all(axis=(1, 3, 5))
    return boxes_from_tiles(t, ndim, blocking_factor, max_grid_size, ratio)


def boxes_from_tiles(t: np.ndarray, ndim: int, blocking_factor: int, max_grid_size: int, ratio: int = 2, parent_align: int = 0,
                     grid_eff: Optional[float] = None, allowed: Optional[np.ndarray] = None) -> List[Box]:
    """t[k, j, i]: tiles (blocking_factor fine cells on a side) to refine -> fine boxes by the library's host functions (shared with the
    C++ host): grid_eff given -> Berger-Rigoutsos clustering + simplify + maxSize (qk_amr_cluster_berger_rigoutsos, the steps of
    amrex::AmrMesh::MakeNewGrids); None -> the round-1 rule, tiles merged greedily (x, then y, then z) up to max_grid_size"""
    import ctypes as C
    assert ratio == 2
    flags = np.ascontiguousarray(t, dtype=np.int32)
    tz, ty, tx = flags.shape
    cap = int(flags.sum()) + 1
    out = (capi.Box * cap)()
    if grid_eff is not None:
        cap = int(flags.size) + 1
        out = (capi.Box * cap)()
        ok = None if allowed is None else np.ascontiguousarray(allowed, dtype=np.int32)
        n = capi.lib().qk_amr_cluster_berger_rigoutsos(flags.ctypes.data_as(C.c_void_p), None if ok is None else ok.ctypes.data_as(C.c_void_p),
                                                       (C.c_int * 3)(tx, ty, tz), ndim, blocking_factor, max_grid_size, float(grid_eff), out, cap)
    else:
        n = capi.lib().qk_amr_cluster_tiles(flags.ctypes.data_as(C.c_void_p), (C.c_int * 3)(tx, ty, tz), ndim, blocking_factor, max_grid_size, parent_align, out, cap)
    if n < 0:
        raise capi.QkError(f"tile clustering failed ({n})")
    return [([out[b].lo[d] for d in range(3)], [out[b].hi[d] for d in range(3)]) for b in range(n)]


def covered_mask(boxes: Sequence[Box], shape) -> np.ndarray:
    m = np.zeros(shape, dtype=bool)
    for lo, hi in boxes:
        m[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = True
    return m


# ------------------------------------------------------------------------------------------------ boxes -> ranks (host)
def max_size(boxes: Sequence[Box], chunk: Sequence[int], blocking_factor: int) -> List[Box]:
    """amrex::BoxArray::maxSize(chunk): every box longer than chunk[d] is cut into ceil(len / chunk) nearly equal pieces (in units of the blocking
    factor, which divides every box edge), x fastest"""
    out: List[Box] = []
    for lo, hi in boxes:
        cuts = []
        for d in range(3):
            n = hi[d] - lo[d] + 1
            if n <= chunk[d]:
                cuts.append([(lo[d], hi[d])])
                continue
            unit = blocking_factor if n % blocking_factor == 0 else 1
            m, nb = n // unit, -(-n // chunk[d])
            base, rem = divmod(m, nb)
            pieces, a = [], lo[d]
            for i in range(nb):
                ln = (base + (1 if i < rem else 0)) * unit
                pieces.append((a, a + ln - 1))
                a += ln
            cuts.append(pieces)
        for kz in cuts[2]:
            for ky in cuts[1]:
                for kx in cuts[0]:
                    out.append(([kx[0], ky[0], kz[0]], [kx[1], ky[1], kz[1]]))
    return out


def chop_grids(boxes: Sequence[Box], target: int, max_grid_size: int, blocking_factor: int, domain_len: Sequence[int], ndim: int = 3) -> List[Box]:
    """amrex::AmrMesh::ChopGrids (refine_grid_layout = 1, AMReX's default, which the reference does not change): while a level has fewer boxes than
    there are ranks, halve the chunk size — the longest direction first — as long as the halved size is a multiple of the blocking factor, and
    re-apply maxSize.  Restated from AMReX's source as documented (AMReX is not vendored): unpinned, like the rest of grid generation."""
    boxes = [(list(lo), list(hi)) for lo, hi in boxes]
    chunk = [min(max_grid_size, domain_len[d]) if d < ndim else 1 for d in range(3)]
    while len(boxes) < target:
        prev = list(chunk)
        for d in sorted(range(ndim), key=lambda a: (-chunk[a], -a)):  # largest chunk first (ties: the highest dimension, as the reversed stable sort)
            new = chunk[d] // 2
            if len(boxes) < target and new > 0 and new % blocking_factor == 0:
                chunk[d] = new
                boxes = max_size(boxes, chunk, blocking_factor)
        if chunk == prev:
            break
    return boxes


def _morton(i: int, j: int, k: int) -> int:
    m = 0
    for bit in range(20):
        m |= ((i >> bit) & 1) << (3 * bit) | ((j >> bit) & 1) << (3 * bit + 1) | ((k >> bit) & 1) << (3 * bit + 2)
    return m


def distribute_sfc(boxes: Sequence[Box], nranks: int, rank_load: Optional[Sequence[int]] = None, unit: int = 1) -> List[int]:
    """amrex::DistributionMapping::SFCProcessorMap as AMReX's default strategy does it: the boxes in Morton order of their low corner are cut
    into nranks contiguous runs of about equal volume (a run that the last box pushed over the mean gives that box back), and the runs are dealt,
    heaviest first, to the ranks in order of how little they already hold (`rank_load`: cells of the coarser levels) — a level with fewer boxes
    than ranks lands on the least loaded ranks.  Restated (AMReX not vendored): unpinned; deterministic and identical on every rank."""
    n = len(boxes)
    if nranks == 1:
        return [0] * n
    vol = [int(np.prod([hi[d] - lo[d] + 1 for d in range(3)])) for lo, hi in boxes]
    order = sorted(range(n), key=lambda b: (_morton(*[boxes[b][0][d] // unit for d in range(3)]), b))
    per = sum(vol) / nranks
    runs: List[List[int]] = []
    K, total = 0, 0.0
    for i in range(nranks):
        run, v = [], 0.0
        while K < n and (i == nranks - 1 or v < per):
            v += vol[order[K]]
            run.append(order[K])
            K += 1
        total += v
        if total / (i + 1) > per and len(run) > 1 and i < nranks - 1:
            K -= 1
            total -= vol[run.pop()]
        runs.append(run)
    load = list(rank_load) if rank_load is not None else [0] * nranks
    ranks = sorted(range(nranks), key=lambda r: (load[r], r))  # LeastUsedCPUs
    heavy = sorted(range(nranks), key=lambda i: (-sum(vol[b] for b in runs[i]), i))
    owner = [0] * n
    for r, i in zip(ranks, heavy):
        for b in runs[i]:
            owner[b] = r
    return owner


def _coarsen(box: Box, r: int = 2) -> Box:
    return [x // r for x in box[0]], [x // r for x in box[1]]


class CoarseShadow:
    """The coarse-level data a refined level needs and produces, on the FINE level's distribution — what AMReX builds inside
    FillPatchTwoLevels (the coarse patch under the fine boxes' ghost cells, filled by ParallelCopy + the coarse physical boundary conditions,
    then interpolated locally), average_down (the coarsened fine MultiFab, then ParallelCopy to the coarse level) and YAFluxRegister (m_cfpatch).
    Boxes: the local fine boxes coarsened; 3 ghost cells (2 under the fine ghost cells + 1 of interpolation stencil).  Only several ranks use
    it: with one rank the plans read and write the parent's arrays directly."""
    NG = 3

    def __init__(self, amr: "AmrSimulation", parent: "AmrLevelSim", child: "AmrLevelSim"):
        ctx, rank = amr.ctx, amr.rank
        self.parent, self.child, self.ncomp = parent, child, child.ncomp_cc
        self.boxes = [_coarsen(b) for b in child.all_boxes]
        self.owner = list(child.owner)
        mine = [b for b, o in zip(self.boxes, self.owner) if o == rank]
        self.lev = Level(ctx, parent.geom.ndim, mine)
        self.old = MultiFab(self.lev, self.ncomp, self.NG, fill=0.0)
        self.new = MultiFab(self.lev, self.ncomp, self.NG, fill=0.0)
        self._key_old = self._key_new = None
        # ghost-cell interpolation reads the three ghost layers and the outermost valid layer of a shadow box, the reflecting boundary conditions of
        # the ghost layers the three outermost valid layers: the rest is a hole in the plan
        holes = [([lo[d] + self.NG for d in range(3)], [hi[d] - self.NG for d in range(3)]) for lo, hi in self.boxes]
        self.from_parent = ParallelCopy(ctx, parent.geom, parent.all_boxes, parent.owner, self.boxes, self.owner, self.ncomp, rank, dst_nghost=self.NG, dst_holes=holes)
        self._from_parent_whole: Optional[ParallelCopy] = None
        self.to_parent = ParallelCopy(ctx, parent.geom, self.boxes, self.owner, parent.all_boxes, parent.owner, self.ncomp, rank)
        # physical boundary conditions of the coarse patch (the cbc of FillPatchTwoLevels): the level-0 kernel over the shadow's ghost slabs
        self.bc = GhostExchange(self.lev, parent.geom, self.ncomp, self.NG, mine, [rank] * len(mine), rank, amr.bcs, amr.dirichlet) if mine else None
        self.interp = InterpFromCoarse(self.lev, child.lev, child.geom, NGHOST_CC, all_fine_boxes=child.all_boxes)
        self.avgdown = AverageDown(self.lev, child.lev)

    def _fill(self, dst: MultiFab, src: MultiFab, plan: ParallelCopy):
        plan(src, dst)
        if self.bc is not None and not self.parent.geom.is_all_periodic():
            c = self.lev.ctx
            c.check(c.L.qk_FillPhysicalBoundary_subset(self.bc.h, c.stream(), dst.ptr, self.bc.bcs, self.bc.dirichlet, capi.BOXES_ALL), "FillPhysicalBoundary(shadow)")

    def ensure(self, which: str):
        """the parent's old / new state under this rank's fine boxes — fetched once per version of the parent's state"""
        p = self.parent
        src = p.state_old_cc_ if which == "old" else p.state_new_cc_
        key = (id(src), p._state_gen)
        if getattr(self, "_key_" + which) != key:
            self._fill(self.old if which == "old" else self.new, src, self.from_parent)
            setattr(self, "_key_" + which, key)
        return self.old if which == "old" else self.new

    def fill_whole_new(self) -> MultiFab:
        """every cell of the grown shadow boxes from the parent's new state (a (re)made level interpolates all of its cells)"""
        if self._from_parent_whole is None:
            p, rank = self.parent, self.parent.amr.rank
            self._from_parent_whole = ParallelCopy(self.lev.ctx, p.geom, p.all_boxes, p.owner, self.boxes, self.owner, self.ncomp, rank, dst_nghost=self.NG)
        self._fill(self.new, self.parent.state_new_cc_, self._from_parent_whole)
        self._key_new = None
        return self.new

    def average_down(self, fine_state: MultiFab, crse_state: MultiFab):
        self.avgdown(fine_state, self.new, 0, self.ncomp)
        self._key_new = None
        self.to_parent(self.new, crse_state)


# ------------------------------------------------------------------------------------------------ one level
class AmrLevelSim(HydroSimulation):
    """HydroSimulation over the boxes of one AMR level: same advance (fused stages, FOFC, retries); the ghost fill of a refined level
    adds the coarse -> fine interpolation, and every successful advance feeds the flux registers."""

    def __init__(self, amr: "AmrSimulation", lev: int, boxes: Sequence[Box], owner: Sequence[int]):
        self.amr, self.ilev = amr, lev
        g0 = amr.geom0
        geom = Geometry(g0.ndim, [g0.n_cell[d] * (2 ** lev if d < g0.ndim else 1) for d in range(3)], list(g0.prob_lo), list(g0.prob_hi), list(g0.periodic))
        self._init_simulation(geom, boxes, owner)
        self.min_overlap_cells = 1 << 62  # the early/late split of the uniform-grid exchange is not combined with the coarse-fine fill
        self.store_flux_rk2 = True
        self.shadow: Optional[CoarseShadow] = None  # several ranks: the parent's data under this level's boxes, on this level's distribution
        self._state_gen = 0  # bumped by every writer of state_new_cc_ / state_old_cc_ (the _new_ghosts_filled setter): versions the shadows' copies
        self.t_old = self.t_new = 0.0
        self._fill_time = 0.0
        self.cf_interp: Optional[InterpFromCoarse] = None
        self.fluxreg: Optional[FluxRegister] = None  # between level lev-1 and this level
        self.avgdown: Optional[AverageDown] = None
        for name in ("cflNumber_", "densityFloor_", "tempFloor_", "reconstructionOrder_", "integratorOrder_", "useDualEnergy_", "abortOnFofcFailure_"):
            setattr(self, name, getattr(amr, name))

    def _init_simulation(self, geom, boxes, owner):
        amr = self.amr
        HydroSimulation.__init__(self, amr.ctx, geom, amr.traits, amr.bcs, None, amr.dirichlet, rank=amr.rank, nranks=amr.nranks, boxes=boxes, owner=owner)

    @property
    def _new_ghosts_filled(self) -> bool:
        return self.__dict__.get("_ngf", False)

    @_new_ghosts_filled.setter
    def _new_ghosts_filled(self, v: bool):  # every writer of the level's states clears the flag: that is also what versions the children's shadows
        self.__dict__["_ngf"] = bool(v)
        if not v:
            self._state_gen = self.__dict__.get("_state_gen", 0) + 1

    def link_to_parent(self, parent: "AmrLevelSim"):
        amr = self.amr
        if amr.nranks == 1:  # the parent's arrays are local: the plans read and write them directly
            self.cf_interp = InterpFromCoarse(parent.lev, self.lev, self.geom, NGHOST_CC)
            self.fluxreg = FluxRegister(parent.lev, self.lev, parent.geom, 6)
            self.avgdown = AverageDown(parent.lev, self.lev)
        else:
            self.shadow = CoarseShadow(amr, parent, self)
            self.cf_interp, self.avgdown = self.shadow.interp, None
            self.fluxreg = DistFluxRegister(amr.ctx, parent.lev, parent.geom, parent.all_boxes, parent.owner, self.lev, self.all_boxes, self.owner,
                                            self.shadow.lev, self.shadow.boxes, 6, amr.rank)
        parent._update_flux_mask(self.fluxreg)

    def _update_flux_mask(self, child_fluxreg: FluxRegister):
        """The carried form on a level with refined children (AmrSimulation.rk2_carry_rhs; level 0 only: the one level whose size makes the form of
        the RK2 average matter).  incrementFluxRegisters reads flux_rk2 on the coarse-fine faces alone: the coarse cells of the register's items are
        marked (qk_hydro_stage_args::flux_mask), the fused stage keeps F1 and writes flux_rk2 on their faces exactly as the reference's form does
        on every face, and every other face — 99.9 % of them — stays carried.  Rebuilt whenever the child level is (re)linked."""
        if not (getattr(self.amr, "rk2_carry_rhs", False) and self.ilev == 0 and self.integratorOrder_ == 2):
            return
        m = MultiFab(self.lev, 1, 1, dtype=torch.int8, fill=0)
        win = {}  # box -> bounding box of its marked cells
        for _d, _side, _fb, cb, lo, hi, sh in child_fluxreg.items():
            beg = m.begins[cb]
            sl = tuple(slice(lo[k] + sh[k] - beg[k], hi[k] + sh[k] - beg[k] + 1) for k in (2, 1, 0))
            m.fabs[cb][0][sl] = 1
            wlo, whi = win.setdefault(cb, ([2 ** 30] * 3, [-2 ** 30] * 3))
            for k in range(3):
                wlo[k], whi[k] = min(wlo[k], lo[k] + sh[k]), max(whi[k], hi[k] + sh[k])
        # the descriptors the kernels see are WINDOWS (include/quokka_amd.h, flux_mask): the bounding box of a box's marked cells, none for a box
        # without — a face outside the window is dropped without reading a byte
        tab = m.host_table
        for b in range(len(m.fabs) if getattr(self.amr, "flux_mask_windows", True) else 0):  # (False: whole-box descriptors — tests)
            beg = [int(x) for x in tab[b]["begin"]]
            wlo, whi = win.get(b, (beg, [x - 1 for x in beg]))
            tab[b]["p"] = int(tab[b]["p"]) + (wlo[0] - beg[0]) + int(tab[b]["jstride"]) * (wlo[1] - beg[1]) + int(tab[b]["kstride"]) * (wlo[2] - beg[2])
            tab[b]["begin"], tab[b]["end"] = wlo, [x + 1 for x in whi]
        m.table = torch.from_numpy(tab.view(np.uint8).reshape(-1)).to(m.table.device)
        m.__dict__.pop("_subtables", None)
        self.flux_mask = m
        self.store_flux_rk2 = False
        self.rk2_carry_rhs = True

    def reflux_from(self, child: "AmrLevelSim"):
        """flux_reg_[lev+1]->Reflux(state_new_cc_[lev]) (reference src/simulation.hpp:1308).  Several ranks (DistFluxRegister): the coarse part
        goes straight into the local state, the fine part travels from the fine boxes' ranks to the owners of the coarse cells (ParallelAdd)."""
        child.fluxreg.Reflux(self.state_new_cc_)

    # --- ghost fill (FillPatchTwoLevels)
    def fillBoundaryConditions(self, state: MultiFab):
        if self.ilev == 0:
            super().fillBoundaryConditions(state)
        else:
            self.ghost.fill(state, before_physbc=lambda: self._interp_from_parent(state, self._fill_time, self.cf_interp))

    def _interp_from_parent(self, state: MultiFab, time: float, plan: InterpFromCoarse):
        p = self.amr.levels[self.ilev - 1]
        t0, t1 = p.t_old, p.t_new
        eps = 1.0e-10 * max(abs(t1 - t0), 1.0e-300)
        sh = self.shadow  # several ranks: the parent's states as this rank's copy under its fine boxes (same values: the same interpolated bits)
        old = (lambda: sh.ensure("old")) if sh is not None else (lambda: p.state_old_cc_)
        new = (lambda: sh.ensure("new")) if sh is not None else (lambda: p.state_new_cc_)
        if abs(time - t1) <= eps or t1 == t0:
            c = new()
            plan(state, c, c, 1.0, 0.0, self.ncomp_cc, self.amr.amrInterpMethod_, True)
        elif abs(time - t0) <= eps:
            c = old()
            plan(state, c, c, 1.0, 0.0, self.ncomp_cc, self.amr.amrInterpMethod_, True)
        else:  # amrex::FillPatch time interpolation: ((t1 - t) old + (t - t0) new) / (t1 - t0)
            plan(state, old(), new(), (t1 - time) / (t1 - t0), (time - t0) / (t1 - t0), self.ncomp_cc, self.amr.amrInterpMethod_, True)

    def _before_fill(self, stage: int, dt: float):
        self._fill_time = self._t_adv + (dt if stage == 2 else 0.0)  # reference src/QuokkaSimulation.hpp:1076, :1204

    def advanceHydroAtLevel(self, state_old_tmp: MultiFab, dt_lev: float, time=None) -> bool:
        ok = super().advanceHydroAtLevel(state_old_tmp, dt_lev, self._t_adv if time is None else time)
        if ok:  # incrementFluxRegisters (reference src/QuokkaSimulation.hpp:1303-1306, src/simulation.hpp:1369-1386)
            amr, l = self.amr, self.ilev
            # integratorOrder_ == 1: the step's fluxes are the stage-1 fluxes (scale 1), which stay in halfFlux
            flux = self.fluxRk2() if self.integratorOrder_ == 2 else self.halfFlux
            if amr.do_reflux and l < amr.finest_level:
                amr.levels[l + 1].fluxreg.CrseAdd(flux, self.geom.dx, dt_lev)
            if amr.do_reflux and l > 0:
                self.fluxreg.FineAdd(flux, self.geom.dx, dt_lev)
        self._t_adv += dt_lev
        return ok

    def advance_level(self, time: float, dt_lev: float) -> bool:
        """advanceSingleTimestepAtLevel: state_new <- advance(state_old = previous state_new), with the retries of
        advanceHydroAtLevelWithRetries (reference src/QuokkaSimulation.hpp:885-990); every attempt restarts at `time`"""
        if self.amr._spec_entries is not None and self._can_defer_verdict():
            return self._advance_level_deferred(time, dt_lev)
        self._signal_of_state_new = None
        self._old_ghosts_filled = False
        self._new_ghosts_filled = False
        self.state_old_cc_, self.state_new_cc_ = self.state_new_cc_, self.state_old_cc_
        amr, l = self.amr, self.ilev
        fr_as_fine = self.fluxreg if (amr.do_reflux and l > 0) else None
        fr_as_crse = amr.levels[l + 1].fluxreg if (amr.do_reflux and l < amr.finest_level) else None
        if fr_as_fine is not None:
            fr_as_fine.save()  # originalFineData (reference src/QuokkaSimulation.hpp:894-900)
        for retry_count in range(7):
            nsubsteps = 2 ** retry_count
            dt_step = dt_lev / nsubsteps
            if retry_count > 0:
                self.counters["retries"] += 1
                # the substeps of the failed attempt that succeeded have already been added to the registers: back to the pre-advance
                # state, or Reflux would count them twice (reference src/QuokkaSimulation.hpp:919-929)
                if fr_as_crse is not None:
                    fr_as_crse.reset()
                if fr_as_fine is not None:
                    fr_as_fine.restore()
            self._t_adv = time
            in_place = nsubsteps == 1 and not self.strang_sources  # (a Strang-split source changes the old state: then always the copy)
            old = self.state_old_cc_ if in_place else self.state_old_tmp
            if not in_place:
                self.state_old_tmp.copy_from(self.state_old_cc_)
            success = True
            for substep in range(nsubsteps):
                if substep > 0:
                    self.state_old_tmp.copy_comps_from(self.state_new_cc_, 0, 6)
                success = self.advanceHydroAtLevel(old, dt_step)
                if not success:
                    break
            if success:
                # the stage-1 fill of an in-place attempt was applied to state_old_cc_ itself (at `time`, from the parent's unchanged states) and
                # nothing writes the old state afterwards: timeStepWithSubcycling need not fill it again for the children (hydro only: the
                # radiation subcycle mirrors its substeps into the old state)
                self._old_ghosts_filled = in_place and not isinstance(self, RadhydroSimulation)
                return True
        return False

    def FixupState(self):
        self._fixup_state(self.state_new_cc_)

    # --- a speculative coarse step (AmrSimulation.overlap_children): verdicts are read once, at its end
    def _can_defer_verdict(self) -> bool:
        return (self.use_fused and self.integratorOrder_ == 2 and self.speculate_stage2 and not self.strang_sources and type(self) is AmrLevelSim
                and len(self.amr._spec_entries) < self.amr._spec_words.shape[0])

    def _advance_level_deferred(self, time: float, dt_lev: float) -> bool:
        """advance_level inside a speculative coarse step: both stages and the register increments are enqueued, the eight words the stages
        report in are copied to a slot of the hierarchy's log, and the step counts as good until AmrSimulation reads the log — a bad verdict there
        rolls the whole coarse step back and redoes it in the ordinary order (first-order flux correction, retries)."""
        amr, l = self.amr, self.ilev
        self._signal_of_state_new = None
        self._new_ghosts_filled = False
        self.state_old_cc_, self.state_new_cc_ = self.state_new_cc_, self.state_old_cc_
        self._t_adv = time
        old, inter, new = self.state_old_cc_, self.state_inter_cc_, self.state_new_cc_
        self._err_latched, self._unfused_ran = False, False
        self._fused_begin(1, both=True)
        self._launch_stage(1, old, old, inter, dt_lev, slot=0)
        self._launch_stage(2, inter, old, new, dt_lev, slot=1)
        k = len(amr._spec_entries)
        amr._spec_words[k].copy_(self._dev_words, non_blocking=True)
        amr._spec_entries.append((self, dt_lev, k))
        self._stage1_left_F1 = not self._carry_active()
        flux = self.fluxRk2()
        if amr.do_reflux and l < amr.finest_level:
            amr.levels[l + 1].fluxreg.CrseAdd(flux, self.geom.dx, dt_lev)
        if amr.do_reflux and l > 0:
            self.fluxreg.FineAdd(flux, self.geom.dx, dt_lev)
        self._t_adv += dt_lev
        self._old_ghosts_filled = True  # (stage 1 filled the old state's ghost cells in place)
        return True

    # --- the children beside the far boxes (AmrSimulation.overlap_children)
    def advance_level_begin(self, time: float, dt_lev: float, near: List[int], far: List[int]):
        """advance_level with the verdict deferred: ghost fill + stage 1 of all boxes, ghost fill + stage 2 of the `near` boxes (the ones the
        children read) on the compute stream, stage 2 of the `far` boxes on a second stream with a scratch array of its own; the physical
        boundaries of the near boxes' new state and the coarse side of the child's flux register follow on the compute stream, so that the
        children can be enqueued at once.  Nothing is read back: advance_level_join() gives the verdict."""
        self._signal_of_state_new = None
        self._old_ghosts_filled = False
        self._new_ghosts_filled = False
        self.state_old_cc_, self.state_new_cc_ = self.state_new_cc_, self.state_old_cc_
        amr = self.amr
        self._t_adv = time
        old, inter, new = self.state_old_cc_, self.state_inter_cc_, self.state_new_cc_
        self._err_latched, self._unfused_ran = False, False
        if getattr(self, "_far_groups_key", None) != (tuple(near), tuple(far)):
            mk = lambda idx: (Level(self.ctx, self.geom.ndim, [self.my_boxes[b] for b in idx]), list(idx))
            self._near_group, self._far_group = mk(near), mk(far)
            # the far boxes as up to 8 launch sets one after the other on the side stream, each holding its share of the wave slots only: the
            # children's small kernels find free slots sooner (QK_AMR_FAR_SPLIT; profiles/round5/ab9_amr_far_split.txt: 1 / 2 / 3 / 7 sets
            # 2217 / 2206 / 2226 / 2253 M on config 5's geometry, 7 far boxes)
            nsplit = max(1, min(len(far), int(__import__("os").environ.get("QK_AMR_FAR_SPLIT", "8"))))
            self._far_parts = [self._far_group] if nsplit == 1 else [mk(far[k::nsplit]) for k in range(nsplit)]
            nbytes = self.ctx.L.qk_hydro_stage_scratch_bytes(self._far_group[0].h, __import__("ctypes").byref(self.traits))
            self._far_scratch = torch.empty(max(nbytes // 8, 1), dtype=torch.float64, device=self.ctx.device)
            # the far boxes fill whatever the children's small kernels leave idle: the LOWEST priority the device offers (the compute stream
            # keeps its own), so that a child kernel is dispatched as soon as it is ready
            import os as _os
            least, greatest = torch.cuda.Stream.priority_range()
            prio = {"low": least, "high": greatest}.get(_os.environ.get("QK_AMR_FAR_PRIORITY", "low"), 0)
            self._far_stream = torch.cuda.Stream(device=self.ctx.device, priority=prio)
            # (Measured and rejected in round 5, profiles/round5/ab8_amr_cu_mask_rejected.txt: the far boxes on a CU-masked stream that leaves every
            # 8th / 4th / 16th compute unit to the children: 2220 -> 1936 / 1937 / 1649 M.)
            for b in range(self.lev.nboxes):  # the physical-boundary slabs of the near boxes first ("local only" subset of the ghost plan)
                self.ghost.set_box_remote(b, b in set(far))
            self._far_groups_key = (tuple(near), tuple(far))
        # Level 0 has no coarse-fine ghost cells: its intermediate state may travel as primitives (qk_hydro_stage_args::prim_out / prim_in, as on
        # a plain level).  A flagged cell in either stage makes advance_level_join() fail: the coarse step is then redone in the ordinary
        # order, which does not use the hand-off.
        self._prim_now = self.ilev == 0 and type(self) is AmrLevelSim and self._prim_handoff_state_ok()
        self._fused_begin(1, both=True)
        self._before_fill(1, dt_lev)
        self.fillBoundaryConditions(old)
        self._fused_launch(1, old, old, inter, dt_lev, slot=0)
        self._before_fill(2, dt_lev)
        self.fillBoundaryConditions(inter)
        main = torch.cuda.current_stream(self.ctx.device)
        ev = torch.cuda.Event()
        ev.record(main)
        self._fused_launch(2, inter, old, new, dt_lev, group=self._near_group, slot=1)
        if __import__("os").environ.get("QK_AMR_FAR_AFTER_NEAR", "0") == "1":  # (experiment: the far boxes start when the near boxes are done)
            ev = torch.cuda.Event()
            ev.record(main)
        with torch.cuda.stream(self._far_stream):
            self._far_stream.wait_event(ev)
            for part in self._far_parts:
                self._fused_launch(2, inter, old, new, dt_lev, group=part, slot=1, scratch=self._far_scratch)
            self._prim_now = False
            # FixupState of the far boxes (reference src/simulation.hpp:1308-1312: after Reflux and AverageDownTo — neither touches a far box, so
            # for these cells it may as well run now, beside the children); its maxima wait in words 4, 5 for the near boxes' (_fixup_near)
            if amr.overlap_fixup:
                c = self.ctx
                import ctypes as C
                c.check(c.L.qk_hydro_FixupState(self._far_group[0].h, c.stream(), C.byref(self.traits), float(self.densityFloor_), float(self.tempFloor_),
                                                int(self.useDualEnergy_), new.subset_ptr(self._far_group[1]), C.c_void_p(self._dev_fix.data_ptr() + 16),
                                                C.c_void_p(self._dev_fix.data_ptr() + 32)), "qk_hydro_FixupState(far)")
            self._far_done = torch.cuda.Event()
            self._far_done.record(self._far_stream)
        # what the children read of the new state beyond the near boxes' valid cells lies beyond the domain (AmrSimulation._overlap_split)
        c = self.ctx
        c.check(c.L.qk_FillPhysicalBoundary_subset(self.ghost.h, c.stream(), new.ptr, self.ghost.bcs, self.ghost.dirichlet, capi.BOXES_LOCAL_ONLY),
                "FillPhysicalBoundary(near)")
        if amr.do_reflux:  # incrementFluxRegisters, coarse side: the register cells lie in the near boxes
            amr.levels[self.ilev + 1].fluxreg.CrseAdd(self.fluxRk2() if self.integratorOrder_ == 2 else self.halfFlux, self.geom.dx, dt_lev)
        self._t_adv += dt_lev
        self._join_dt = dt_lev

    def _fixup_near(self):
        """FixupState of the near boxes after Reflux and AverageDownTo; with the far boxes' (advance_level_begin) the whole level has had it"""
        import ctypes as C
        c = self.ctx
        c.check(c.L.qk_hydro_FixupState(self._near_group[0].h, c.stream(), C.byref(self.traits), float(self.densityFloor_), float(self.tempFloor_),
                                        int(self.useDualEnergy_), self.state_new_cc_.subset_ptr(self._near_group[1]), C.c_void_p(self._dev_fix.data_ptr() + 16),
                                        C.c_void_p(self._dev_fix.data_ptr())), "qk_hydro_FixupState(near)")
        self._signal_of_state_new = None
        self._fix_words_pending, self._fix_far_words, self._fix_error_pending = True, True, True

    def advance_level_join(self) -> bool:
        """the verdict of advance_level_begin: both stages clean on every box, no error flag, no CFL violation"""
        torch.cuda.current_stream(self.ctx.device).wait_event(self._far_done)
        vals = self._read_words()
        ok = self._fused_end(1, 0, vals) == 0 and self._fused_end(2, 1, vals) == 0
        if self._err_latched:
            raise capi.QkError("density is negative in SyncDualEnergy! abort!! (reference src/hydro/hydro_system.hpp:834-836)")
        if ok:
            self._stage1_left_F1 = not self._carry_active()
            ok = not self.isCflViolated(self._join_dt)
        return ok


class RadAmrLevelSim(AmrLevelSim, RadhydroSimulation):
    """One AMR level of a radiation-hydrodynamics run: the hydro advance of AmrLevelSim, then the radiation subcycle of RadhydroSimulation
    (reference src/QuokkaSimulation.hpp:653-707) with ghost cells interpolated from the parent at the substep's time and the radiation fluxes
    of both stages added to the flux registers with weight dt_radiation / 2 (:1726-1860, expandFluxArrays: the register of the radiation block)."""

    def _init_simulation(self, geom, boxes, owner):
        amr = self.amr
        RadhydroSimulation.__init__(self, amr.ctx, geom, amr.traits, amr.rad_traits, amr.bcs, None, rank=amr.rank, nranks=amr.nranks, dirichlet=amr.dirichlet,
                                    boxes=boxes, owner=owner)
        self.fluxreg_rad: Optional[FluxRegister] = None
        self.store_rad_flux = True  # the flux registers read the face fluxes of both stages
        self.is_hydro_enabled = amr.is_hydro_enabled
        for name in ("radiationCflNumber_", "maxSubsteps_", "radiationReconstructionOrder_"):
            setattr(self, name, getattr(amr, name))
        self._rad_time = 0.0

    def link_to_parent(self, parent: "AmrLevelSim"):
        AmrLevelSim.link_to_parent(self, parent)
        amr = self.amr
        if amr.nranks == 1:
            self.fluxreg_rad = FluxRegister(parent.lev, self.lev, parent.geom, self.nrad)
            self.fluxreg_rad.set_state_component(RAD0)
        else:  # the register of the radiation block, distributed like the hydro one (register component n <-> state component RAD0 + n)
            self.fluxreg_rad = DistFluxRegister(amr.ctx, parent.lev, parent.geom, parent.all_boxes, parent.owner, self.lev, self.all_boxes, self.owner,
                                                self.shadow.lev, self.shadow.boxes, self.nrad, amr.rank, state_comp0=RAD0)

    def reflux_from(self, child: "AmrLevelSim"):
        AmrLevelSim.reflux_from(self, child)
        child.fluxreg_rad.Reflux(self.state_new_cc_)

    def _rad_registers(self, flux, dt_radiation: float):
        amr, l = self.amr, self.ilev
        if amr.do_reflux and l < amr.finest_level:
            amr.levels[l + 1].fluxreg_rad.CrseAdd(flux, self.geom.dx, 0.5 * dt_radiation)
        if amr.do_reflux and l > 0:
            self.fluxreg_rad.FineAdd(flux, self.geom.dx, 0.5 * dt_radiation)

    def advanceRadiationForwardEuler(self, dt_radiation: float):
        self._fill_time = self._rad_time  # :1743 fillBoundaryConditions(state_old, ..., time)
        RadhydroSimulation.advanceRadiationForwardEuler(self, dt_radiation)
        self._rad_registers(self.radFluxOld, dt_radiation)

    def advanceRadiationMidpointRK2(self, dt_radiation: float):
        self._fill_time = self._rad_time + dt_radiation  # :1764 (time + dt_radiation)
        RadhydroSimulation.advanceRadiationMidpointRK2(self, dt_radiation)
        self._rad_registers(self.radFlux, dt_radiation)
        self._rad_time += dt_radiation

    def advance_level(self, time: float, dt_lev: float) -> bool:
        if self.is_hydro_enabled:
            if not AmrLevelSim.advance_level(self, time, dt_lev):
                return False
        else:  # :681-685: the hydro variables are carried over
            self._signal_of_state_new = None
            self._old_ghosts_filled = self._new_ghosts_filled = False
            self.state_old_cc_, self.state_new_cc_ = self.state_new_cc_, self.state_old_cc_
            self.state_new_cc_.copy_comps_from(self.state_old_cc_, 0, RAD0)
        self._rad_time = time
        return self.subcycleRadiationAtLevel(time, dt_lev)

    def FixupState(self):
        if self.is_hydro_enabled:
            AmrLevelSim.FixupState(self)


# ------------------------------------------------------------------------------------------------ the hierarchy
class AmrSimulation:
    def __init__(self, ctx: Context, geom0: Geometry, traits: capi.HydroTraits, bcs, max_level: int, max_grid_size: int = 128, blocking_factor: int = 32,
                 n_error_buf: int = 3, regrid_int: int = 2, dirichlet=None, rank: int = 0, nranks: int = 1, cluster_within_parent: Optional[bool] = None,
                 rad_traits: Optional[capi.RadTraits] = None):
        assert geom0.ndim == 3, "the AMR driver runs the fused 3-D path"
        self.ctx, self.geom0, self.traits, self.bcs, self.dirichlet = ctx, geom0, traits, bcs, dirichlet
        # radiation on every level (reference src/QuokkaSimulation.hpp:653-707, :1577-1722): the levels are RadAmrLevelSim
        self.rad_traits = rad_traits
        self.is_hydro_enabled = True
        self.rad_source: Optional[Callable] = None  # SetRadEnergySource of the problem, per level geometry
        self.radiationCflNumber_, self.maxSubsteps_, self.radiationReconstructionOrder_ = 0.3, 10, 3
        self.rank, self.nranks = rank, nranks
        # Several ranks: every level has its own box -> rank map (distribute_sfc over the level's boxes, least loaded ranks first), and a level
        # with fewer boxes than ranks is chopped until every rank owns one (chop_grids: AMReX's refine_grid_layout) — the grids are clustered
        # globally, as with one rank.  `refine_grid_layout_target`: the box count to chop for (None: the number of ranks; tests give a one-rank
        # run the target of the several-rank run it is compared with).  cluster_within_parent = True keeps round 3's rule for comparison runs
        # (grids clustered inside each level-0 box).
        self.cluster_within_parent = False if cluster_within_parent is None else cluster_within_parent
        self.refine_grid_layout_target: Optional[int] = None
        # level 0: "bricks" (compact, cheapest exchange) or "interleaved" (rank = Morton index of the box mod nranks)
        self.level0_distribution = "bricks"
        self.max_level, self.max_grid_size, self.blocking_factor = max_level, max_grid_size, blocking_factor
        self.n_error_buf, self.regrid_int = n_error_buf, regrid_int
        # amr.grid_eff (reference tests/blast_amr_maxlev2.in: 0.7): Berger-Rigoutsos clustering as amrex::AmrMesh::MakeNewGrids does it.
        # clustering = "tiles": the round-1 rule (every flagged tile refined, greedy merge) — more refined cells, no efficiency parameter
        self.grid_eff, self.clustering = 0.7, "berger_rigoutsos"
        self.do_reflux, self.do_subcycle = True, True
        # One rank, hydro: while the boxes of a level that no child reads ("far") finish their second stage on a side stream, the children
        # advance on the compute stream — chains of small latency-bound kernels beside a launch that fills the GPU.  The level's verdict
        # (redo counts, CFL check) then arrives after the children have run: they run speculatively and are rolled back if it is bad
        # (_snapshot_above / _restore_above; the level is redone the ordinary way).  Same kernels on the same data: same bits.
        self.overlap_children = False
        self.overlap_fixup = True  # (with overlap_children: FixupState of the far boxes on the side stream as well)
        self.overlap_max_fine_fraction = 0.25  # speculate only while the finer levels are small (their snapshot and their kernels)
        self.overlap_stats = {"overlapped": 0, "rolled_back": 0}
        self.defer_child_verdicts = True  # (with overlap_children: one device -> host read per coarse step instead of one per level step)
        self._spec_entries: Optional[list] = None  # inside a speculative coarse step: (level, dt, slot of _spec_words) per deferred level step
        self._spec_words = None  # (32 slots of eight words on the device, made on first use)
        self.amrInterpMethod_ = 1
        self.cflNumber_, self.densityFloor_, self.tempFloor_ = 0.3, 0.0, 0.0
        self.reconstructionOrder_, self.integratorOrder_, self.useDualEnergy_, self.abortOnFofcFailure_ = 3, 2, 1, 1
        self.stopTime_, self.maxTimesteps_ = 1.0, 10 ** 9
        self.levels: List[AmrLevelSim] = []
        self.istep = [0] * (max_level + 1)
        self.last_regrid_step = [0] * (max_level + 1)
        self.dt_ = [1.0e100] * (max_level + 1)
        self.tNew_ = 0.0
        self.cellUpdates_ = 0
        self.cellUpdatesEachLevel_ = [0] * (max_level + 1)
        self.ErrorEst: Optional[Callable[["AmrSimulation", int, MultiFab], None]] = None  # (amr, lev, tags) -> sets tags on level lev
        self.initial_conditions: Optional[Callable] = None  # fn(geom_of_level) -> fn(i, j, k) -> conserved state on index grids
        self.static_fine_boxes: Optional[List[List[Box]]] = None  # [lev-1] -> boxes of level lev: fixed grids instead of ErrorEst

    @property
    def finest_level(self) -> int:
        return len(self.levels) - 1

    def CountCells(self, lev: int) -> int:
        return self.levels[lev].CountCells()  # all boxes of the level (every rank counts the global work, as the reference does)

    def _ensure_new_ghosts(self, l: int):
        """ghost cells of level l's new state at its new time — filled unless they are current: timeStepWithSubcycling fills level l for its children
        just before the children's regrid runs, and regrid(0) tags two levels that both need level 0 (one whole-level fill per coarse step and a
        half).  Every writer of state_new_cc_ clears the flag (advance_level, reflux / AverageDownTo / FixupState, the end of a regrid)."""
        L = self.levels[l]
        if getattr(L, "_children_beside_far_boxes", False):
            return  # (its far boxes are still in flight; what the children read of it is complete: _overlap_split)
        if not getattr(L, "_new_ghosts_filled", False):
            L._fill_time = L.t_new
            L.fillBoundaryConditions(L.state_new_cc_)
            L._new_ghosts_filled = True

    # ------------------------------------------------------------------ hierarchy construction
    def _tags_on_level(self, lev: int) -> np.ndarray:
        from .amr import TagBoxArray
        for l in range(lev + 1):  # ghost cells of every level up to lev (a refined level interpolates from its parent's ghost-filled state)
            self._ensure_new_ghosts(l)
        L = self.levels[lev]
        tags = TagBoxArray(L.lev)
        self.ErrorEst(self, lev, tags)
        n = L.geom.n_cell
        dense = np.zeros((n[2], n[1], n[0]), dtype=bool)
        for b, (lo, hi) in enumerate(L.my_boxes):
            dense[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = tags.fab_numpy(b)[0] == capi.TAG_SET
        return dense

    def _tile_flags(self, lev: int) -> np.ndarray:
        """ErrorEst on level lev -> tags buffered by n_error_buf -> one flag per blocking-factor tile (qk_amr_tile_flags: only the tile
        flags leave the device)"""
        import ctypes as C
        from .amr import TagBoxArray
        for l in range(lev + 1):
            self._ensure_new_ghosts(l)
        L = self.levels[lev]
        tags = TagBoxArray(L.lev)
        self.ErrorEst(self, lev, tags)
        n, tile = L.geom.n_cell, self.blocking_factor // 2
        nt = [n[d] // tile for d in range(3)]
        flags = np.zeros((nt[2], nt[1], nt[0]), dtype=np.int32)
        dom = capi.Box((C.c_int * 3)(0, 0, 0), (C.c_int * 3)(n[0] - 1, n[1] - 1, n[2] - 1))
        per = (C.c_int * 3)(*[int(x) for x in self.geom0.periodic])
        self.ctx.check(self.ctx.L.qk_amr_tile_flags_periodic(L.lev.h, self.ctx.stream(), tags.ptr, C.byref(dom), per, self.n_error_buf, tile,
                                                             flags.ctypes.data_as(C.c_void_p)), "qk_amr_tile_flags_periodic")
        if self.nranks > 1:  # every rank clusters the same global flags
            import torch.distributed as dist
            from . import comm
            t = torch.from_numpy(flags).to(self.ctx.device)
            comm.all_reduce(t, dist.ReduceOp.MAX)
            flags = t.cpu().numpy()
        return flags != 0

    def _new_grids(self, lev: int, finer_boxes: Optional[List[Box]], base: Optional[int] = None) -> List[Box]:
        """boxes of level lev+1 from the tags on level lev (+ the cells under an already chosen level lev+2, buffered: proper nesting).
        `base`: the level whose regrid this is (AmrCore::regrid(base)): levels above it are being rebuilt together, finest first, so
        the nesting domain of level lev+1 comes from the grids of `base` (which stay), not from the old level-lev grids — level lev
        is rebuilt afterwards around the new level lev+1.  Otherwise the finest level could never leave the old extent of its parent."""
        if self.static_fine_boxes is not None:
            return self.static_fine_boxes[lev] if lev < len(self.static_fine_boxes) else []
        base = lev if base is None else base
        L = self.levels[lev]
        tile = self.blocking_factor // 2  # tile edge in level-lev cells
        assert tile >= 4 and all(L.geom.n_cell[d] % tile == 0 for d in range(3)), "blocking_factor must be >= 8 and divide the domain"
        t = self._tile_flags(lev)
        tz, ty, tx = t.shape
        if finer_boxes:  # level lev+2 boxes: their level-lev footprint grown by 2 cells (ghost reach + stencil of level lev+1) must be refined
            nt3 = (tx, ty, tz)
            for lo, hi in finer_boxes:  # (through a periodic face the footprint continues on the other side of the domain)
                idx = []
                for d in range(3):
                    a, b = (lo[d] // 4 - 2) // tile, (hi[d] // 4 + 2) // tile
                    if not self.geom0.periodic[d]:
                        a, b = max(a, 0), min(b, nt3[d] - 1)
                    idx.append(np.arange(a, b + 1) % nt3[d])
                t[np.ix_(idx[2], idx[1], idx[0])] = True
        allowed = np.ones_like(t)
        if base > 0:  # proper nesting: a tile and its 26 neighbours (>= 4 cells: ghost reach 2 + stencil 1) lie on cells of `base` (refined) or beyond the domain
            r = 2 ** (lev - base)
            cov = np.ones((tz + 2, ty + 2, tx + 2), dtype=bool)
            cov[1:-1, 1:-1, 1:-1] = False
            for lo, hi in self.levels[base].all_boxes:
                cov[lo[2] * r // tile + 1:(hi[2] * r + r - 1) // tile + 2, lo[1] * r // tile + 1:(hi[1] * r + r - 1) // tile + 2,
                    lo[0] * r // tile + 1:(hi[0] * r + r - 1) // tile + 2] = True
            # Levels base+1 .. lev are rebuilt in the same regrid, each nested in the next coarser one with a margin of one of ITS tiles: seen
            # from level lev the grids of `base` shrink by 2^(lev-base+1) - 2 tiles before the usual one-tile check.  Beyond a periodic face
            # lies the other side of the domain (which `base` need not cover); beyond a physical one nothing that interpolation would read:
            # only those border tiles stay "covered".
            per = self.geom0.periodic
            for _ in range(2 ** (lev - base + 1) - 1):
                if per[0]:
                    cov[:, :, 0], cov[:, :, -1] = cov[:, :, -2].copy(), cov[:, :, 1].copy()
                if per[1]:
                    cov[:, 0, :], cov[:, -1, :] = cov[:, -2, :].copy(), cov[:, 1, :].copy()
                if per[2]:
                    cov[0, :, :], cov[-1, :, :] = cov[-2, :, :].copy(), cov[1, :, :].copy()
                inner = ~dilate(~cov, 1, 3)[1:-1, 1:-1, 1:-1]
                cov[1:-1, 1:-1, 1:-1] = inner
            allowed = cov[1:-1, 1:-1, 1:-1].copy()
            t &= allowed
        eff = self.grid_eff if self.clustering == "berger_rigoutsos" else None
        if not self.cluster_within_parent:
            return boxes_from_tiles(t, 3, self.blocking_factor, self.max_grid_size, 2, grid_eff=eff, allowed=allowed)
        boxes: List[Box] = []
        s = 2 ** lev  # level-0 boxes in level-lev index space: every level is clustered inside its level-0 ancestors (one rank each)
        for lo0, hi0 in self.levels[0].all_boxes:  # the level-0 boxes of ALL ranks, in the same order everywhere
            lo, hi = [x * s for x in lo0], [x * s + s - 1 for x in hi0]
            a = [lo[d] // tile for d in range(3)]
            b = [hi[d] // tile for d in range(3)]
            sub = t[a[2]:b[2] + 1, a[1]:b[1] + 1, a[0]:b[0] + 1]
            for blo, bhi in boxes_from_tiles(sub, 3, self.blocking_factor, self.max_grid_size, 2, grid_eff=eff,
                                             allowed=allowed[a[2]:b[2] + 1, a[1]:b[1] + 1, a[0]:b[0] + 1]):
                boxes.append(([blo[d] + 2 * lo[d] for d in range(3)], [bhi[d] + 2 * lo[d] for d in range(3)]))
        return boxes

    def _chop(self, lev: int, boxes: List[Box]) -> List[Box]:
        """AmrMesh::ChopGrids: a level with fewer boxes than ranks is cut until every rank can own one (as far as the blocking factor allows)"""
        target = self.nranks if self.refine_grid_layout_target is None else self.refine_grid_layout_target
        if target <= 1 or len(boxes) >= target or not boxes:
            return boxes
        g0 = self.geom0
        return chop_grids(boxes, target, self.max_grid_size, self.blocking_factor, [g0.n_cell[d] * 2 ** lev for d in range(3)], g0.ndim)

    def _owners_of(self, lev: int, boxes: List[Box]) -> List[int]:
        """box -> rank map of a refined level: space-filling curve over the level's boxes, the least loaded ranks (cells of the coarser levels) first"""
        if self.nranks == 1:
            return [0] * len(boxes)
        load = [0] * self.nranks
        for L in self.levels[:lev]:
            for (lo, hi), o in zip(L.all_boxes, L.owner):
                load[o] += int(np.prod([hi[d] - lo[d] + 1 for d in range(3)]))
        return distribute_sfc(boxes, self.nranks, load, unit=self.blocking_factor)

    def _make_level(self, lev: int, boxes: List[Box]) -> AmrLevelSim:
        if lev == 0:
            from .simulation import distribute_boxes, distribute_boxes_interleaved
            fn = distribute_boxes_interleaved if self.level0_distribution == "interleaved" else distribute_boxes
            lattice = int(np.prod([-(-self.geom0.n_cell[d] // self.max_grid_size) for d in range(self.geom0.ndim)]))
            if self.nranks == 1:
                owner = [0] * len(boxes)
            elif len(boxes) == lattice:
                owner = fn(boxes, self.nranks, self.geom0.n_cell, [self.max_grid_size] * 3)
            else:  # level 0 was chopped below max_grid_size (fewer boxes than ranks): no box lattice to cut into bricks
                owner = distribute_sfc(boxes, self.nranks, None, unit=self.blocking_factor)
        else:
            owner = self._owners_of(lev, boxes)  # (`boxes` are chopped by the caller: _chop)
        L = (RadAmrLevelSim if self.rad_traits is not None else AmrLevelSim)(self, lev, boxes, owner)
        if self.rad_traits is not None and self.rad_source is not None:
            L.SetRadEnergySource = self.rad_source(L.geom)  # fn(geom of the level) -> fn(i, j, k, time)
        if lev > 0:
            L.link_to_parent(self.levels[lev - 1])
        return L

    def setInitialConditions(self):
        """AmrCore::InitFromScratch: level 0, then finer levels from the tags of the initial conditions (MakeNewLevelFromScratch uses
        the problem's initial conditions on every level, reference src/simulation.hpp:1656-1702), then AverageDown"""
        g0 = self.geom0
        self.levels = [self._make_level(0, self._chop(0, chop_domain(g0.n_cell, [self.max_grid_size] * 3)))]
        self._set_level_ic(0)
        for lev in range(self.max_level):
            boxes = self._chop(lev + 1, self._new_grids(lev, None))
            if not boxes:
                break
            self.levels.append(self._make_level(lev + 1, boxes))
            self._set_level_ic(lev + 1)
        # AmrMesh::MakeNewGrids(time) iterates at start-up: once a level exists, the levels below it are rebuilt around it (top-down, with
        # the nesting footprints), which may make room for one more level — a level whose first grids were too narrow to hold a child
        # (the ring around a sharp pulse) gets its child in the next pass.  Every (re)built level takes the problem's initial conditions.
        if self.static_fine_boxes is None:
            for _ in range(self.max_level + 1):
                before = [sorted(map(str, L.all_boxes)) for L in self.levels]
                self.regrid(0)
                for lev in range(1, self.finest_level + 1):
                    self._set_level_ic(lev)
                if [sorted(map(str, L.all_boxes)) for L in self.levels] == before:
                    break
        for lev in range(self.finest_level - 1, -1, -1):
            self.AverageDownTo(lev)

    def _set_level_ic(self, lev: int):
        L = self.levels[lev]
        fn = self.initial_conditions(L.geom)
        for b, (lo, hi) in enumerate(L.my_boxes):
            k, j, i = np.meshgrid(np.arange(lo[2], hi[2] + 1), np.arange(lo[1], hi[1] + 1), np.arange(lo[0], hi[0] + 1), indexing="ij")
            L.state_new_cc_.valid(b).copy_(torch.from_numpy(np.ascontiguousarray(fn(i, j, k))))
        L._new_ghosts_filled = False
        L.state_old_cc_.copy_from(L.state_new_cc_)
        L.t_old = L.t_new = self.tNew_

    def regrid(self, base: int):
        """amrex::AmrCore::regrid(base, time): new grids for levels base+1 .. ; data from the old level where it exists, interpolated from
        the coarser level elsewhere (RemakeLevel / MakeNewLevelFromCoarse, reference src/simulation.hpp:1656-1702)"""
        if self.static_fine_boxes is not None:
            return
        new_boxes: dict = {}
        top = min(self.finest_level + 1, self.max_level)
        finer = None
        for lev in range(top - 1, base - 1, -1):  # finest first, so that coarser levels can enclose the finer ones
            if lev > self.finest_level:
                continue
            boxes = self._new_grids(lev, finer if lev + 2 <= self.max_level else None, base)
            new_boxes[lev + 1] = boxes
            finer = boxes if boxes else None
        new_boxes = {lev: self._chop(lev, boxes) for lev, boxes in new_boxes.items()}
        for lev in range(base + 1, self.max_level + 1):
            boxes = new_boxes.get(lev, [])
            if not boxes:
                del self.levels[lev:]
                break
            old = self.levels[lev] if lev <= self.finest_level else None
            if old is not None and sorted(map(str, old.all_boxes)) == sorted(map(str, [(list(lo), list(hi)) for lo, hi in boxes])):
                continue
            new = self._make_level(lev, boxes)
            parent = self.levels[lev - 1]
            if new.shadow is None:
                parent._fill_time = parent.t_new
                parent.fillBoundaryConditions(parent.state_new_cc_)
                whole = InterpFromCoarse(parent.lev, new.lev, new.geom, NGHOST_CC, whole_fab=True)
                whole(new.state_new_cc_, parent.state_new_cc_, parent.state_new_cc_, 1.0, 0.0, new.ncomp_cc, self.amrInterpMethod_, True)
                if old is not None:
                    _copy_overlap(old.state_new_cc_, old.my_boxes, new.state_new_cc_, new.my_boxes)
            else:  # several ranks: the parent's cells under the new boxes arrive in the shadow; the old level's cells from their owners
                c = new.shadow.fill_whole_new()
                whole = InterpFromCoarse(new.shadow.lev, new.lev, new.geom, NGHOST_CC, whole_fab=True)
                whole(new.state_new_cc_, c, c, 1.0, 0.0, new.ncomp_cc, self.amrInterpMethod_, True)
                if old is not None:
                    ParallelCopy(self.ctx, new.geom, old.all_boxes, old.owner, new.all_boxes, new.owner, new.ncomp_cc, self.rank)(old.state_new_cc_, new.state_new_cc_)
            new.state_old_cc_.copy_from(new.state_new_cc_)
            new.t_old, new.t_new = parent.t_new, parent.t_new
            new.dt_ = old.dt_ if old is not None else 1.0e100
            if lev <= self.finest_level:
                self.levels[lev] = new
            else:
                self.levels.append(new)
            # the child of a remade level needs new inter-level plans — unless it is about to be remade (or removed) itself: its OLD grids need
            # not lie inside the new parent (a shrinking hierarchy)
            nxt = new_boxes.get(lev + 1, [])
            if lev + 1 <= self.finest_level and nxt and sorted(map(str, self.levels[lev + 1].all_boxes)) == sorted(map(str, [(list(lo), list(hi)) for lo, hi in nxt])):
                self.levels[lev + 1].link_to_parent(new)
        for k in range(base, self.finest_level + 1):  # reference src/simulation.hpp:1257-1259
            self.levels[k].FixupState()
            self.levels[k]._new_ghosts_filled = False

    # ------------------------------------------------------------------ inter-level operators
    def AverageDownTo(self, crse_lev: int):
        f = self.levels[crse_lev + 1]
        if f.shadow is None:
            f.avgdown(f.state_new_cc_, self.levels[crse_lev].state_new_cc_, 0, f.ncomp_cc)
        else:  # averaged on the fine boxes' ranks, then to the owners of the coarse cells
            f.shadow.average_down(f.state_new_cc_, self.levels[crse_lev].state_new_cc_)
        self.levels[crse_lev]._new_ghosts_filled = False

    # ------------------------------------------------------------------ time stepping
    def computeTimestep(self):
        """reference src/simulation.hpp:744-818 with do_subcycle = 1: dt_0 = min_l (n_factor_l * dt_cfl_l), growth <= 1.1x, dt_l = dt_0 / 2^l"""
        dt_tmp = [self.levels[l].computeTimestepAtLevel() for l in range(self.finest_level + 1)]
        dt_0, n_factor = dt_tmp[0], 1
        for l in range(self.finest_level + 1):
            if l > 0:
                n_factor *= 2
            dt_tmp[l] = min(dt_tmp[l], 1.1 * self.dt_[l])
            dt_0 = min(dt_0, n_factor * dt_tmp[l])
        eps = 1.0e-3 * dt_0
        if self.tNew_ + dt_0 > self.stopTime_ - eps:
            dt_0 = self.stopTime_ - self.tNew_
        self.dt_[0] = dt_0
        for l in range(1, self.max_level + 1):
            self.dt_[l] = self.dt_[l - 1] / 2.0

    # ------------------------------------------------------------------ children beside the far boxes
    def _overlap_split(self, lev: int):
        """(near, far) local boxes of level lev if its children can be advanced beside the second stage of the far boxes, else None.
        near: every box the child level reads — the coarse cells under its ghost-cell interpolation (stencil included), the register cells of
        its flux register, the cells it averages down to.  Required of every interpolation item: the coarse cells it reads lie in the VALID
        region of its coarse box or beyond a physical boundary (never in ghost cells another box fills — those wait for the far boxes)."""
        L = self.levels[lev]
        if not (self.overlap_children and lev == 0 and self.nranks == 1 and self.rad_traits is None and lev < self.finest_level and self.do_reflux
                and L.use_fused and L.integratorOrder_ == 2 and L.speculate_stage2 and not L.strang_sources and L.lev.nboxes > 1):
            return None
        fine_cells = sum(self.CountCells(l) for l in range(lev + 1, self.finest_level + 1))
        if fine_cells > self.overlap_max_fine_fraction * self.CountCells(lev):
            return None
        child = self.levels[lev + 1]
        key = (id(L), id(child), id(child.cf_interp))
        cached = self.__dict__.get("_split_cache")
        if cached is not None and cached[0] == key:
            return cached[1]
        near, ok = set(), True
        per, dom = L.geom.periodic, L.geom.n_cell
        for fb, cb, lo, hi in child.cf_interp.items():
            near.add(cb)
            vlo, vhi = L.my_boxes[cb]
            for d in range(3):
                clo, chi = lo[d] // 2 - 1, hi[d] // 2 + 1
                if clo < vlo[d] and not (vlo[d] == 0 and not per[d]):
                    ok = False
                if chi > vhi[d] and not (vhi[d] == dom[d] - 1 and not per[d]):
                    ok = False
        for _d, _side, _fb, cb, _lo, _hi, _sh in child.fluxreg.items():
            near.add(cb)
        for (flo, fhi) in child.all_boxes:  # the cells AverageDown writes
            for b, (vlo, vhi) in enumerate(L.my_boxes):
                if all(flo[d] // 2 <= vhi[d] and fhi[d] // 2 >= vlo[d] for d in range(3)):
                    near.add(b)
        far = [b for b in range(L.lev.nboxes) if b not in near]
        out = (sorted(near), far) if ok and far and near else None
        self._split_cache = (key, out)
        return out

    def _deferred_verdicts(self, entries) -> bool:
        """the verdicts of the level steps a speculative coarse step deferred (AmrLevelSim._advance_level_deferred): both stages clean, no error
        flag, no CFL violation — one device -> host copy for all of them.  A level whose state nothing has changed since (the finest one) takes
        the CFL maxima of its last step, as the ordinary advance leaves them."""
        if not entries:
            return True
        n = len(entries)
        h = self._spec_words[:n].cpu()
        f = h.view(torch.float64)
        last = {}
        for L, dt, k in entries:
            nbad1, nbad2 = int(h[k, 2]), int(h[k, 6])
            err = (int(h[k, 3]) | int(h[k, 7])) & 0xFFFFFFFF
            sig0, sig1 = float(f[k, 4]), float(f[k, 5])
            if nbad1 != 0 or nbad2 != 0:
                self.overlap_stats.setdefault("reasons", []).append(f"level {L.ilev}: {nbad1} / {nbad2} cells flagged for the first-order flux correction")
                return False
            if err != 0:
                raise capi.QkError("density is negative in SyncDualEnergy! abort!! (reference src/hydro/hydro_system.hpp:834-836)")
            if dt > 1.1 * (L.cflNumber_ * (L.min_dx() / sig0)):  # isCflViolated
                self.overlap_stats.setdefault("reasons", []).append(f"level {L.ilev}: CFL violated")
                return False
            last[id(L)] = (L, sig0, sig1)
        for L, sig0, sig1 in last.values():
            if L in self.levels and not L._fix_words_pending and L._signal_of_state_new is None:
                L._signal_of_state_new = (sig0, sig1)
        return True

    def _snapshot_above(self, lev: int):
        per = []
        for L in self.levels[lev + 1:]:
            per.append((L, L.state_new_cc_, L.state_old_cc_, L.state_new_cc_.storage.clone(), L.t_old, L.t_new, L.dt_, dict(L.counters), L.istep))
        return {"levels": list(self.levels), "istep": list(self.istep), "last": list(self.last_regrid_step), "cu": self.cellUpdates_,
                "cul": list(self.cellUpdatesEachLevel_), "per": per}

    def _restore_above(self, lev: int, snap):
        self.levels = list(snap["levels"])
        self.istep, self.last_regrid_step = list(snap["istep"]), list(snap["last"])
        self.cellUpdates_, self.cellUpdatesEachLevel_ = snap["cu"], list(snap["cul"])
        for L, new, old, data, t_old, t_new, dt, counters, istep in snap["per"]:
            L.state_new_cc_, L.state_old_cc_ = new, old
            new.storage.copy_(data)
            L.t_old, L.t_new, L.dt_, L.counters, L.istep = t_old, t_new, dt, counters, istep
            L._signal_of_state_new = None
            L._old_ghosts_filled = False
            L._new_ghosts_filled = False
            # whatever FixupState flagged on this level belongs to the discarded attempt (a deferred child stage is not corrected: its negative
            # densities reach FixupState through AverageDownTo): the sticky error word and the pending reads go with it
            L._dev_fix[2:3].zero_()
            L._fix_error_pending = False
            L._fix_far_words = False
            L._fix_words_pending = False
        for l in range(lev + 1, self.finest_level + 1):  # plans a regrid of the discarded attempt may have re-pointed
            self.levels[l].link_to_parent(self.levels[l - 1])
        self.__dict__.pop("_split_cache", None)

    def timeStepWithSubcycling(self, lev: int, time: float):
        if self.regrid_int > 0 and lev < self.max_level and self.istep[lev] > self.last_regrid_step[lev] and self.istep[lev] % self.regrid_int == 0:
            self.regrid(lev)
            for k in range(lev, self.finest_level + 1):
                self.last_regrid_step[k] = self.istep[k]
        L = self.levels[lev]
        L.t_old = L.t_new
        L.t_new = L.t_new + self.dt_[lev]
        if self.do_reflux and lev < self.finest_level:
            self.levels[lev + 1].fluxreg.reset()
            if self.rad_traits is not None:
                self.levels[lev + 1].fluxreg_rad.reset()
        split = self._overlap_split(lev)
        if split is not None:
            snap = self._snapshot_above(lev)
            L.advance_level_begin(time, self.dt_[lev], *split)
            L._children_beside_far_boxes = True
            if self.defer_child_verdicts and self._spec_words is None:
                self._spec_words = torch.zeros(32, 8, dtype=torch.int64, device=self.ctx.device)
            self._spec_entries = [] if self.defer_child_verdicts else None
            try:
                for i in range(2):
                    self.timeStepWithSubcycling(lev + 1, time + i * self.dt_[lev + 1])
                entries = self._spec_entries
            finally:
                L._children_beside_far_boxes = False
                self._spec_entries = None
            ok = L.advance_level_join()
            if not ok:
                self.overlap_stats.setdefault("reasons", []).append(f"level {lev}: verdict of the level itself")
            ok = ok and self._deferred_verdicts(entries)
            if getattr(self, "_force_speculation_failure", False):  # (tests: the rollback path)
                self._force_speculation_failure, ok = False, False
            if ok:
                self.overlap_stats["overlapped"] += 1
                self.istep[lev] += 1
                self.cellUpdates_ += self.CountCells(lev)
                self.cellUpdatesEachLevel_[lev] += self.CountCells(lev)
                if self.do_reflux:
                    L.reflux_from(self.levels[lev + 1])
                self.AverageDownTo(lev)
                if self.overlap_fixup:
                    L._fixup_near()
                else:
                    L.FixupState()
                L._new_ghosts_filled = False
                return
            # the level's step was not clean: the children advanced on a state that will not stand.  Back to the start of the step, then the
            # ordinary order (first-order flux correction / retries, then the children).
            self.overlap_stats["rolled_back"] += 1
            L._dev_fix[2:3].zero_()  # (whatever FixupState of the far boxes flagged belongs to the discarded attempt)
            self._restore_above(lev, snap)
            L = self.levels[lev]
            L.state_old_cc_, L.state_new_cc_ = L.state_new_cc_, L.state_old_cc_
            L._signal_of_state_new = None
            self.levels[lev + 1].fluxreg.reset()
        if not L.advance_level(time, self.dt_[lev]):
            raise capi.QkError(f"QUOKKA FATAL ERROR: Hydro update exceeded max_retries on level {lev}")
        self.istep[lev] += 1
        self.cellUpdates_ += self.CountCells(lev)
        self.cellUpdatesEachLevel_[lev] += self.CountCells(lev)
        if lev < self.finest_level:
            # the children interpolate their ghost cells from this level's old and new states: both need their own ghost cells (one rank; with
            # several the children's shadows take the valid cells and apply the boundary conditions themselves)
            for st, t in ((L.state_old_cc_, L.t_old), (L.state_new_cc_, L.t_new)) if self.nranks == 1 else ():
                if st is L.state_old_cc_ and getattr(L, "_old_ghosts_filled", False):
                    continue
                L._fill_time = t
                L.fillBoundaryConditions(st)
                if st is L.state_new_cc_:
                    L._new_ghosts_filled = True
            for i in range(2):
                if lev < self.finest_level:
                    self.timeStepWithSubcycling(lev + 1, time + i * self.dt_[lev + 1])
            if lev < self.finest_level:
                if self.do_reflux:
                    L.reflux_from(self.levels[lev + 1])
                self.AverageDownTo(lev)
                L.FixupState()
                L._new_ghosts_filled = False

    def step(self):
        self.computeTimestep()
        self.timeStepWithSubcycling(0, self.tNew_)
        self.tNew_ += self.dt_[0]

    def evolve(self):
        while self.istep[0] < self.maxTimesteps_ and self.tNew_ < self.stopTime_:
            self.step()
            if self.tNew_ >= self.stopTime_ - 1.0e-6 * self.dt_[0]:
                break

    # ------------------------------------------------------------------ diagnostics
    def use_carried_form(self, on: bool = True):
        """hydro.rk2_carry_rhs for the hierarchy: level 0 advances in the carried form of the RK2 average and forms flux_rk2 only on its coarse-fine
        faces (AmrLevelSim._update_flux_mask); refined levels — small, and read by both neighbours' registers — keep the reference's form"""
        self.rk2_carry_rhs = bool(on)
        L0 = self.levels[0]
        if on and self.finest_level >= 1:
            L0._update_flux_mask(self.levels[1].fluxreg)
        elif not on:
            L0.flux_mask, L0.store_flux_rk2, L0.rk2_carry_rhs = None, True, False

    def composite_sum(self, comp: int) -> float:
        """volume integral of a conserved component over the composite grid (cells under a finer level are not counted): every level's
        own sum minus, for each (local box, box of the next level) pair, the part of the local box that the finer box covers — the
        boxes of a level are disjoint, so nothing is subtracted twice.  Device reductions, one number per box pair to the host."""
        total = 0.0
        for l, L in enumerate(self.levels):
            vol = L.geom.dx[0] * L.geom.dx[1] * L.geom.dx[2]
            finer = [([x // 2 for x in lo], [x // 2 for x in hi]) for lo, hi in self.levels[l + 1].all_boxes] if l < self.finest_level else []
            for b, (lo, hi) in enumerate(L.my_boxes):
                v = L.state_new_cc_.valid(b)[comp]
                s = float(v.sum(dtype=torch.float64))
                for flo, fhi in finer:
                    a = [max(lo[d], flo[d]) for d in range(3)]
                    e = [min(hi[d], fhi[d]) for d in range(3)]
                    if all(a[d] <= e[d] for d in range(3)):
                        s -= float(v[a[2] - lo[2]:e[2] - lo[2] + 1, a[1] - lo[1]:e[1] - lo[1] + 1, a[0] - lo[0]:e[0] - lo[0] + 1].sum(dtype=torch.float64))
                total += s * vol
        if self.nranks > 1:
            import torch.distributed as dist
            from . import comm
            t = torch.tensor([total], dtype=torch.float64, device=self.ctx.device)
            comm.all_reduce(t, dist.ReduceOp.SUM)
            total = float(t.item())
        return total


def _copy_overlap(src: MultiFab, src_boxes, dst: MultiFab, dst_boxes):
    """valid cells of the old level that the new level still covers (RemakeLevel: FillPatch copies the fine data where it exists)"""
    for sb, (slo, shi) in enumerate(src_boxes):
        for db, (dlo, dhi) in enumerate(dst_boxes):
            lo = [max(slo[d], dlo[d]) for d in range(3)]
            hi = [min(shi[d], dhi[d]) for d in range(3)]
            if any(lo[d] > hi[d] for d in range(3)):
                continue
            s0, d0 = src.begins[sb], dst.begins[db]
            dst.fabs[db][:, lo[2] - d0[2]:hi[2] - d0[2] + 1, lo[1] - d0[1]:hi[1] - d0[1] + 1, lo[0] - d0[0]:hi[0] - d0[0] + 1] = \
                src.fabs[sb][:, lo[2] - s0[2]:hi[2] - s0[2] + 1, lo[1] - s0[1]:hi[1] - s0[1] + 1, lo[0] - s0[0]:hi[0] - s0[0] + 1]


def sedov_amr_problem(ctx: Context, n: int, max_level: int, max_grid_size: int = 128, blocking_factor: int = 32, static_fine_boxes=None, rank: int = 0,
                      nranks: int = 1, cluster_within_parent=None, level0_distribution: str = "bricks", refine_grid_layout_target=None) -> AmrSimulation:
    """reference src/problems/HydroBlast3D/test_hydro3d_blast.cpp + tests/blast_amr_maxlev2.in (BASELINE config 5)"""
    geom = Geometry(3, [n, n, n], [0.0, 0.0, 0.0], [1.2, 1.2, 1.2], [0, 0, 0])
    bcs = []
    for c in range(6):
        lo = [capi.BC_REFLECT_ODD if c == 1 + d else capi.BC_REFLECT_EVEN for d in range(3)]
        bcs.append((lo, list(lo)))
    amr = AmrSimulation(ctx, geom, capi.traits(1.4, False, 3), bcs, max_level, max_grid_size, blocking_factor, rank=rank, nranks=nranks,
                        cluster_within_parent=cluster_within_parent)
    amr.level0_distribution = level0_distribution
    amr.refine_grid_layout_target = refine_grid_layout_target
    amr.static_fine_boxes = static_fine_boxes
    E_blast = 0.851072 / 8.0

    def ic_for(geom_l: Geometry):
        cell_vol = geom_l.dx[0] * geom_l.dx[1] * geom_l.dx[2]

        def ic(i, j, k):
            U = np.zeros((6,) + i.shape)
            U[0] = 1.0
            U[4] = np.where((i == 0) & (j == 0) & (k == 0), E_blast / cell_vol, 1.0e-10 * (E_blast / cell_vol))
            return U
        return ic

    def error_est(a: AmrSimulation, lev: int, tags: MultiFab):  # test_hydro3d_blast.cpp:118-151
        from .amr import tag_relative_gradient
        L = a.levels[lev]
        tag_relative_gradient(L.lev, L.traits, L.state_new_cc_, tags, capi.TAGFIELD_PRESSURE, 0.1, 1.0e-3, False)

    amr.initial_conditions, amr.ErrorEst = ic_for, error_est
    amr.setInitialConditions()
    return amr


def rad_pulse_amr_problem(ctx: Context, n: int, max_level: int, max_grid_size: int = 32, blocking_factor: int = 8, static_fine_boxes=None,
                          hydro: bool = False, kappa: float = 4.0, rank: int = 0, nranks: int = 1, cluster_within_parent=None,
                          tag_threshold: float = 1.5, refine_grid_layout_target=None) -> AmrSimulation:
    """A radiation pulse in a periodic box of gas at rest (units c = c_hat = a_rad = k_B = mu = 1, rho = 1, T = 1, kappa constant): a Gaussian excess
    of radiation energy at the centre spreads across the coarse-fine interfaces and heats the gas — the radiation operators, the exchange and the
    radiation flux registers on every level.  Not a reference problem: a property test (see tests/test_amr_radiation_gpu.py)."""
    geom = Geometry(3, [n, n, n], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [1, 1, 1])
    bcs = [([capi.BC_INT_DIR] * 3, [capi.BC_INT_DIR] * 3) for _ in range(10)]
    traits = capi.traits(5.0 / 3.0, True, 3, mean_molecular_weight=1.0, boltzmann_constant=1.0)
    rt = capi.RadTraits(1.0, 1.0, 1.0, 1.0e-10, 1 if hydro else 0, 0, kappa, kappa, kappa, 1, 0)
    amr = AmrSimulation(ctx, geom, traits, bcs, max_level, max_grid_size, blocking_factor, rad_traits=rt, rank=rank, nranks=nranks,
                        cluster_within_parent=cluster_within_parent)
    amr.is_hydro_enabled = hydro
    amr.refine_grid_layout_target = refine_grid_layout_target
    amr.static_fine_boxes = static_fine_boxes
    amr.radiationCflNumber_ = 0.3

    def ic_for(geom_l: Geometry):
        dx = geom_l.dx

        def ic(i, j, k):
            x, y, z = (i + 0.5) * dx[0] - 0.5, (j + 0.5) * dx[1] - 0.5, (k + 0.5) * dx[2] - 0.5
            U = np.zeros((10,) + i.shape)
            U[0] = 1.0
            U[4] = U[5] = 1.5  # rho c_v T, c_v = 1 / (gamma - 1) / mu
            U[6] = 1.0 + 3.0 * np.exp(-(x * x + y * y + z * z) / (2.0 * 0.08 ** 2))
            return U
        return ic

    def error_est(a: AmrSimulation, lev: int, tags: MultiFab):  # (only used without static grids) radiation energy above tag_threshold
        L = a.levels[lev]
        for b in range(L.lev.nboxes):
            tags.valid(b)[0][L.state_new_cc_.valid(b)[6] > a.tag_threshold] = capi.TAG_SET

    amr.tag_threshold = tag_threshold

    amr.initial_conditions, amr.ErrorEst = ic_for, error_est
    amr.setInitialConditions()
    return amr


def shell_amr_problem(ctx: Context, n: int, max_level: int, table, max_grid_size: int = 128, blocking_factor: int = 32, pow_mode: int = 0) -> AmrSimulation:
    """reference src/problems/RadhydroShell/test_radhydro_shell.cpp + tests/radhydro_shell_amr.in: the radiation-driven shell on a refined
    hierarchy (tags: relative density jump to a neighbour > 0.1 where rho >= 0.01 rho_0, :321-357)"""
    from .radhydro import ShellConstants as S, shell_functions, shell_settings
    geom = Geometry(3, [n, n, n], [0.0, 0.0, 0.0], [S.L_box] * 3, [1, 1, 1])
    bcs = [([capi.BC_INT_DIR] * 3, [capi.BC_INT_DIR] * 3) for _ in range(10)]
    traits = capi.traits(S.gamma_gas, False, 3, mean_molecular_weight=2.2 * capi.M_U, boltzmann_constant=capi.K_B)
    rt = capi.RadTraits(S.c, S.chat, S.a_rad, 0.0, 1, 0, S.kappa0, S.kappa0, S.kappa0, pow_mode)
    amr = AmrSimulation(ctx, geom, traits, bcs, max_level, max_grid_size, blocking_factor, rad_traits=rt)
    for name, value in shell_settings().items():
        setattr(amr, name, value)
    amr.initial_conditions = lambda g: shell_functions(g, table)[0]
    amr.rad_source = lambda g: shell_functions(g, table)[1]

    def error_est(a: AmrSimulation, lev: int, tags: MultiFab):
        L, g = a.levels[lev], NGHOST_CC
        for b in range(L.lev.nboxes):
            rho = L.state_new_cc_.fabs[b][0]
            c = rho[g:-g, g:-g, g:-g]
            d = torch.zeros_like(c)
            for ax in range(3):
                sl = lambda o: tuple(slice(g + (o if a_ == ax else 0), rho.shape[a_] - g + (o if a_ == ax else 0)) for a_ in range(3))
                d = torch.maximum(d, torch.maximum((rho[sl(1)] - c).abs(), (c - rho[sl(-1)]).abs()))
            tags.valid(b)[0][((d / c) > 0.1) & (c >= 1.0e-2 * S.rho_0)] = capi.TAG_SET

    amr.ErrorEst = error_est
    amr.setInitialConditions()
    return amr
