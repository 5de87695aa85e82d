"""Device containers mirroring the AMReX objects on the hot path (BoxArray -> Level, MultiFab,
iMultiFab).  PyTorch is used only as the device allocator / stream provider; the data layout is
the amrex::MultiFab one (per box: Fortran order, component outermost, ghost cells included) and
the descriptor table handed to the C-ABI is an array of amrex::Array4-compatible structs.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import capi

_A4_DTYPE = np.dtype([("p", np.uint64), ("jstride", np.int64), ("kstride", np.int64), ("nstride", np.int64),
                      ("begin", np.int32, 3), ("end", np.int32, 3), ("ncomp", np.int32)], align=True)
assert _A4_DTYPE.itemsize == C.sizeof(capi.Array4) == 64


class Context:
    """qk_ctx: one per process / GPU."""

    def __init__(self, device: int = 0):
        if not torch.cuda.is_available():
            raise capi.QkError("no GPU visible: the quokka_amd hot path has no CPU fallback")
        self.L = capi.lib()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        rc = self.L.qk_ctx_create(C.byref(h), device)
        if rc != capi.QK_OK:
            raise capi.QkError(f"qk_ctx_create failed ({rc})")
        self.h = h

    def stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def check(self, rc: int, what: str = ""):
        capi.check(self.h, rc, what)

    def __del__(self):
        try:
            self.L.qk_ctx_destroy(self.h)
        except Exception:
            pass


class PlanningContext:
    """qk_ctx with QK_DEVICE_HOST_PLANNING: box / ghost-plan logic on the host, no kernels.  Only the multi-rank CPU
    (gloo) tests of the exchange protocol use it; the product path always runs on a Context."""

    def __init__(self):
        self.L = capi.lib()
        self.device = torch.device("cpu")
        h = C.c_void_p()
        rc = self.L.qk_ctx_create(C.byref(h), -1)
        if rc != capi.QK_OK:
            raise capi.QkError(f"qk_ctx_create(planning) failed ({rc})")
        self.h = h

    def stream(self):
        return None

    def check(self, rc: int, what: str = ""):
        capi.check(self.h, rc, what)

    def __del__(self):
        try:
            self.L.qk_ctx_destroy(self.h)
        except Exception:
            pass


def _box_struct(lo, hi) -> capi.Box:
    return capi.Box((C.c_int * 3)(*[int(x) for x in lo]), (C.c_int * 3)(*[int(x) for x in hi]))


class Level:
    """qk_level: the valid (cell-centred) boxes of one AMR level owned by this rank."""

    def __init__(self, ctx: Context, ndim: int, boxes: Sequence[Sequence[Sequence[int]]]):
        self.ctx = ctx
        self.ndim = ndim
        self.boxes = [([int(x) for x in lo], [int(x) for x in hi]) for lo, hi in boxes]
        arr = (capi.Box * len(self.boxes))(*[_box_struct(lo, hi) for lo, hi in self.boxes])
        h = C.c_void_p()
        ctx.check(ctx.L.qk_level_create(ctx.h, C.byref(h), ndim, len(self.boxes), arr), "qk_level_create")
        self.h = h

    @property
    def nboxes(self) -> int:
        return len(self.boxes)

    def num_cells(self) -> int:
        return sum(int(np.prod([hi[d] - lo[d] + 1 for d in range(3)])) for lo, hi in self.boxes)

    def __del__(self):
        try:
            self.ctx.L.qk_level_destroy(self.h)
        except Exception:
            pass


class MultiFab:
    """amrex::MultiFab (dtype float64) / iMultiFab (dtype int32) on the GPU.

    facedir = -1: cell-centred; 0/1/2: nodal in that direction (amrex::convert(ba, e_d)).
    One contiguous allocation holds all boxes (288 GB of HBM: size for few, large allocations).
    """

    def __init__(self, level: Level, ncomp: int, nghost: int, facedir: int = -1, dtype=torch.float64, fill: Optional[float] = None,
                 align_rows: Optional[bool] = None):
        self.level, self.ncomp, self.nghost, self.facedir, self.dtype = level, ncomp, nghost, facedir, dtype
        ctx = level.ctx
        # Row pitch (cell-centred FP64 arrays; QK_ROW_ALIGN=0 restores dense rows for A/B runs): rows 16-double multiples apart, every fab offset
        # so that the first VALID cell of every row of every component begins a 128-byte line.  A dense row of a 128-cell box with 4 ghost
        # cells is 8.5 lines long and starts 32 bytes into a line — the marching sweeps' 512-byte wave accesses then straddle 5 lines and share
        # lines with neighbouring rows and chunks (csrc/qk_hydro_fused.hip: scratchGeom uses the same layout for the scratch arrays).  Any stride
        # is legal in a qk_array4 / amrex::Array4.
        if align_rows is None:
            align_rows = os.environ.get("QK_ROW_ALIGN", "1") in ("1", "2")  # (2: state arrays only, 3: the library's scratch arrays only)
        self.aligned = bool(align_rows) and facedir == -1 and dtype == torch.float64
        self.shapes, self.begins, self.pitches, offsets, total = [], [], [], [], 0
        for lo, hi in level.boxes:
            n = [hi[d] - lo[d] + 1 + (1 if d == facedir else 0) + (2 * nghost if d < level.ndim else 0) for d in range(3)]
            pitch = (n[0] + 15) // 16 * 16 if self.aligned else n[0]  # (a second pad line per row, measured: the pre-pass 10 % slower)
            if self.aligned:
                total = (total + 15) // 16 * 16 + (16 - nghost) % 16
            self.shapes.append((ncomp, n[2], n[1], n[0]))
            self.pitches.append(pitch)
            self.begins.append([lo[d] - (nghost if d < level.ndim else 0) for d in range(3)])
            offsets.append(total)
            total += ncomp * pitch * n[1] * n[2]
        self.offsets = offsets
        self.storage = torch.empty(max(total, 1), dtype=dtype, device=ctx.device)  # (never a NULL pointer, also for a rank without boxes)
        if fill is not None:
            self.storage.fill_(fill)
        elif self.aligned:
            self.storage.zero_()  # pad columns are copied along with whole-storage copies: keep them finite
        self.fabs: List[torch.Tensor] = [
            torch.as_strided(self.storage, s, (p * s[2] * s[1], p * s[2], p, 1), o) for o, s, p in zip(offsets, self.shapes, self.pitches)]
        tab = np.zeros(max(level.nboxes, 1), dtype=_A4_DTYPE)
        for b, (fab, shp, beg, pitch) in enumerate(zip(self.fabs, self.shapes, self.begins, self.pitches)):
            nx, ny, nz = shp[3], shp[2], shp[1]
            tab[b]["p"] = fab.data_ptr()
            tab[b]["jstride"], tab[b]["kstride"], tab[b]["nstride"] = pitch, pitch * ny, pitch * ny * nz
            tab[b]["begin"] = beg
            tab[b]["end"] = [beg[0] + nx, beg[1] + ny, beg[2] + nz]
            tab[b]["ncomp"] = ncomp
        self.host_table = tab
        self.table = torch.from_numpy(tab.view(np.uint8).reshape(-1)).to(ctx.device)

    def comp_span(self, b: int, c0: int, c1: int) -> torch.Tensor:
        """components [c0, c1) of fab b as ONE contiguous run of the storage (pad columns included): the fast way to copy whole components
        between MultiFabs of the same shape (a sliced fab of an aligned MultiFab is a strided view: its copy_ is an element-wise gather)"""
        shp, p = self.shapes[b], self.pitches[b]
        n = p * shp[2] * shp[1]
        return self.storage[self.offsets[b] + c0 * n: self.offsets[b] + c1 * n]

    def copy_comps_from(self, other: "MultiFab", c0: int, c1: int):
        """amrex::MultiFab::Copy(dst, src, c0, c0, c1 - c0, nghost) for two MultiFabs of the same BoxArray, ghost width and layout"""
        assert self.shapes == other.shapes and self.pitches == other.pitches
        if c0 == 0 and c1 == self.ncomp == other.ncomp:
            self.storage.copy_(other.storage)
            return
        for b in range(len(self.shapes)):
            self.comp_span(b, c0, c1).copy_(other.comp_span(b, c0, c1))

    @property
    def ptr(self) -> C.c_void_p:
        """device pointer to the qk_array4[nboxes] table (== MultiFab::arrays())"""
        return C.c_void_p(self.table.data_ptr())

    def subset_ptr(self, boxes: Sequence[int]) -> C.c_void_p:
        """descriptor table of a subset of the boxes (same memory), for launches over a sub-level"""
        key = tuple(boxes)
        cache = self.__dict__.setdefault("_subtables", {})
        if key not in cache:
            cache[key] = self.table.view(-1, 64)[torch.tensor(list(key), dtype=torch.long, device=self.table.device)].contiguous()
        return C.c_void_p(cache[key].data_ptr())

    def valid_slices(self, b: int):
        ng, nd = self.nghost, self.level.ndim
        return tuple([slice(None)] + [slice(ng, -ng) if (d < nd and ng > 0) else slice(None) for d in (2, 1, 0)])

    def valid(self, b: int) -> torch.Tensor:
        return self.fabs[b][self.valid_slices(b)]

    def set_fab(self, b: int, a: np.ndarray):
        assert tuple(a.shape) == tuple(self.shapes[b]), (a.shape, self.shapes[b])
        self.fabs[b].copy_(torch.from_numpy(np.ascontiguousarray(a)).to(self.dtype))

    def fab_numpy(self, b: int) -> np.ndarray:
        return self.fabs[b].contiguous().cpu().numpy()

    def copy_from(self, other: "MultiFab"):
        self.storage.copy_(other.storage)
