"""On-disk formats of the reference: AMReX plotfiles (`HyperCLaw-V1.1` Header + Level_<l>/Cell_H + Cell_D_*)
and Quokka checkpoints (reference src/simulation.hpp:2564-2834), host side.

Readers for both (what `fcompare` and the reference's checkpoint_restart_test.sh look at), and writers for the Python drivers
(`HydroSimulation`, `AmrSimulation`), rank aware: every rank writes the fabs it owns into its own `Cell_D_<rank>` file, rank 0 writes
the headers — the offsets follow from the box sizes alone, so no communication is needed.  Layouts follow
quokka_amd/host/quokka_io.hpp (the C++ mirror's writer), which cites the reference's own header writers.
"""
from __future__ import annotations

import os
import re
import time as _time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

REAL_DESCRIPTOR = "((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))"
Box = Tuple[List[int], List[int]]


# ------------------------------------------------------------------------------------------------ text pieces
def _box_str(lo: Sequence[int], hi: Sequence[int], ndim: int) -> str:
    v = lambda a: "(" + ",".join(str(int(x)) for x in a[:ndim]) + ")"
    return f"({v(lo)} {v(hi)} {v([0, 0, 0])})"


_BOX_RE = re.compile(r"\(\(([-\d,]+)\) \(([-\d,]+)\) \(([-\d,]+)\)\)")


def _parse_box(text: str) -> Box:
    m = _BOX_RE.search(text)
    if m is None:
        raise ValueError(f"not a Box: {text!r}")
    lo = [int(x) for x in m.group(1).split(",")]
    hi = [int(x) for x in m.group(2).split(",")]
    while len(lo) < 3:
        lo.append(0)
        hi.append(0)
    return lo, hi


def _boxarray_str(boxes: Sequence[Box], ndim: int) -> str:
    return f"({len(boxes)} 0\n" + "".join(_box_str(lo, hi, ndim) + "\n" for lo, hi in boxes) + ")"


def _fmt17(x: float) -> str:
    """std::ostream << double with precision(17) (the %.17g form)"""
    return "%.17g" % x


def _prebuild(name: str, nlevels: int, rank: int, barrier=None):
    if rank == 0:
        if os.path.exists(name):
            os.rename(name, f"{name}.old.{int(_time.time() * 1e6) % 10000000:07d}")
        for l in range(nlevels):
            os.makedirs(os.path.join(name, f"Level_{l}"))
    if barrier is not None:
        barrier()


# ------------------------------------------------------------------------------------------------ VisMF
@dataclass
class VisMF:
    ncomp: int
    nghost: int
    boxes: List[Box]
    fabboxes: List[Box] = field(default_factory=list)
    fabs: List[np.ndarray] = field(default_factory=list)  # [comp, k, j, i] over the stored box
    minima: Optional[np.ndarray] = None
    maxima: Optional[np.ndarray] = None


def read_vismf(prefix: str) -> VisMF:
    """amrex::VisMF::Read"""
    with open(prefix + "_H") as f:
        lines = f.read().split("\n")
    vers, how, ncomp = int(lines[0]), int(lines[1]), int(lines[2])
    assert vers == 1 and how == 1, (vers, how)
    nghost = int(lines[3]) if not lines[3].startswith("(") else int(lines[3].strip("()").split(",")[0])
    nboxes = int(lines[4].lstrip("(").split()[0])
    boxes = [_parse_box(lines[5 + b]) for b in range(nboxes)]
    pos = 5 + nboxes
    assert lines[pos].strip() == ")", lines[pos]
    nfod = int(lines[pos + 1])
    fod = []
    for n in range(nfod):
        tag, name, head = lines[pos + 2 + n].split()
        assert tag == "FabOnDisk:"
        fod.append((name, int(head)))
    pos += 2 + nfod
    mm = []
    rest = [ln for ln in lines[pos:] if ln.strip()]
    i = 0
    while i < len(rest) and len(mm) < 2:
        nrow, ncol = (int(x) for x in rest[i].split(","))
        mm.append(np.array([[float(x) for x in rest[i + 1 + r].rstrip(",").split(",")] for r in range(nrow)]).reshape(nrow, ncol))
        i += 1 + nrow
    out = VisMF(ncomp, nghost, boxes, minima=mm[0] if mm else None, maxima=mm[1] if len(mm) > 1 else None)
    d = os.path.dirname(prefix)
    for name, head in fod:
        with open(os.path.join(d, name), "rb") as f:
            f.seek(head)
            hdr = f.readline().decode()
            assert hdr.startswith("FAB " + REAL_DESCRIPTOR), hdr
            lo, hi = _parse_box(hdr[len("FAB " + REAL_DESCRIPTOR):])
            nc = int(hdr.rsplit(" ", 1)[1])
            assert nc == ncomp
            shape = (nc, hi[2] - lo[2] + 1, hi[1] - lo[1] + 1, hi[0] - lo[0] + 1)
            a = np.frombuffer(f.read(8 * int(np.prod(shape))), dtype="<f8").reshape(shape)
        out.fabboxes.append((lo, hi))
        out.fabs.append(a)
    return out


def _vismf_layout(boxes: Sequence[Box], owner: Sequence[int], ncomp: int, nghost: int, ndim: int):
    """(file name, offset, header string, stored box) of every fab: rank r's fabs follow each other in Cell_D_<r>"""
    cursor: Dict[int, int] = {}
    out = []
    for (lo, hi), r in zip(boxes, owner):
        flo = [lo[d] - (nghost if d < ndim else 0) for d in range(3)]
        fhi = [hi[d] + (nghost if d < ndim else 0) for d in range(3)]
        hdr = f"FAB {REAL_DESCRIPTOR}{_box_str(flo, fhi, ndim)} {ncomp}\n"
        npts = int(np.prod([fhi[d] - flo[d] + 1 for d in range(3)]))
        off = cursor.get(r, 0)
        cursor[r] = off + len(hdr) + 8 * npts * ncomp
        out.append((f"Cell_D_{r:05d}", off, hdr, (flo, fhi)))
    return out


def write_vismf(prefix: str, boxes: Sequence[Box], owner: Sequence[int], rank: int, local_fabs: Sequence[np.ndarray], ncomp: int, nghost: int, ndim: int,
                all_reduce_minmax=None, slice_layout: bool = False):
    """amrex::VisMF::Write.  boxes/owner: the whole level; local_fabs: this rank's fabs ([comp, k, j, i] over the box grown by nghost)
    in the order of its boxes.  all_reduce_minmax(min_table, max_table) completes the per-fab tables across ranks (None: one rank).
    slice_layout: the header as the reference's own 2-D slice writer prints it (src/io/DiagFramePlane.cpp:517-572, Write2DMFHeader) — identical
    except for a blank after the closing parenthesis of the box list (:554); tests/test_plotfile_reference_layout.py types that file by hand."""
    layout = _vismf_layout(boxes, owner, ncomp, nghost, ndim)
    mins = np.full((len(boxes), ncomp), np.inf)
    maxs = np.full((len(boxes), ncomp), -np.inf)
    mine = [g for g, r in enumerate(owner) if r == rank]
    assert len(mine) == len(local_fabs)
    if mine:
        with open(f"{os.path.dirname(prefix)}/{layout[mine[0]][0]}", "wb") as f:
            for g, a in zip(mine, local_fabs):
                name, off, hdr, (flo, fhi) = layout[g]
                assert f.tell() == off and a.shape == (ncomp, fhi[2] - flo[2] + 1, fhi[1] - flo[1] + 1, fhi[0] - flo[0] + 1), (a.shape, flo, fhi)
                f.write(hdr.encode())
                f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
                mins[g] = a.reshape(ncomp, -1).min(axis=1)
                maxs[g] = a.reshape(ncomp, -1).max(axis=1)
    if all_reduce_minmax is not None:
        mins, maxs = all_reduce_minmax(mins, maxs)
    if rank != 0:
        return
    with open(prefix + "_H", "w") as f:
        f.write(f"1\n1\n{ncomp}\n{nghost}\n")
        f.write(_boxarray_str(boxes, ndim) + (" \n" if slice_layout else "\n"))
        f.write(f"{len(boxes)}\n")
        for name, off, _, _ in layout:
            f.write(f"FabOnDisk: {name} {off}\n")
        f.write("\n")
        for k, table in enumerate((mins, maxs)):
            f.write(f"{len(boxes)},{ncomp}\n")
            for row in table:
                f.write("".join("%.16e," % v for v in row) + "\n")
            if k == 0:
                f.write("\n")


# ------------------------------------------------------------------------------------------------ plotfiles
@dataclass
class Plotfile:
    varnames: List[str]
    ndim: int
    time: float
    finest_level: int
    prob_lo: List[float]
    prob_hi: List[float]
    ref_ratio: List[int]
    domains: List[Box]
    level_steps: List[int]
    dx: List[List[float]]
    levels: List[VisMF]

    def component(self, name: str, level: int = 0) -> List[np.ndarray]:
        n = self.varnames.index(name)
        return [a[n] for a in self.levels[level].fabs]


def read_plotfile(name: str) -> Plotfile:
    with open(os.path.join(name, "Header")) as f:
        L = f.read().split("\n")
    assert L[0] == "HyperCLaw-V1.1", L[0]
    nvar = int(L[1])
    varnames = L[2:2 + nvar]
    p = 2 + nvar
    ndim, time, finest = int(L[p]), float(L[p + 1]), int(L[p + 2])
    prob_lo = [float(x) for x in L[p + 3].split()]
    prob_hi = [float(x) for x in L[p + 4].split()]
    ref_ratio = [int(x) for x in L[p + 5].split()]
    domains = [_parse_box(m.group(0)) for m in _BOX_RE.finditer(L[p + 6])]
    steps = [int(x) for x in L[p + 7].split()]
    dx = [[float(x) for x in L[p + 8 + l].split()] for l in range(finest + 1)]
    p += 8 + finest + 1
    assert int(L[p]) == 0 and int(L[p + 1]) == 0  # cartesian, no boundary width
    p += 2
    levels = []
    for l in range(finest + 1):
        lev, ngrids, _t = L[p].split()
        assert int(lev) == l
        p += 2 + int(ngrids) * ndim
        levels.append(read_vismf(os.path.join(name, L[p])))
        assert len(levels[-1].boxes) == int(ngrids)
        p += 1
    return Plotfile(varnames, ndim, time, finest, prob_lo, prob_hi, ref_ratio, domains, steps, dx, levels)


def write_plotfile_header(name: str, varnames: Sequence[str], ndim: int, time: float, prob_lo, prob_hi, domains: Sequence[Box], level_steps: Sequence[int],
                          dx: Sequence[Sequence[float]], level_boxes: Sequence[Sequence[Box]], ref_ratio: int = 2):
    finest = len(level_boxes) - 1
    with open(os.path.join(name, "Header"), "w") as f:
        f.write("HyperCLaw-V1.1\n")
        f.write(f"{len(varnames)}\n" + "".join(v + "\n" for v in varnames))
        f.write(f"{ndim}\n{_fmt17(time)}\n{finest}\n")
        f.write("".join(_fmt17(prob_lo[d]) + " " for d in range(ndim)) + "\n")
        f.write("".join(_fmt17(prob_hi[d]) + " " for d in range(ndim)) + "\n")
        f.write("".join(f"{ref_ratio} " for _ in range(finest)) + "\n")
        f.write("".join(_box_str(lo, hi, ndim) + " " for lo, hi in domains) + "\n")
        f.write("".join(f"{s} " for s in level_steps) + "\n")
        for l in range(finest + 1):
            f.write("".join(_fmt17(dx[l][d]) + " " for d in range(ndim)) + "\n")
        f.write("0\n0\n")
        for l in range(finest + 1):
            f.write(f"{l} {len(level_boxes[l])} {_fmt17(time)}\n{level_steps[l]}\n")
            dlo = domains[l][0]
            for lo, hi in level_boxes[l]:
                for d in range(ndim):
                    f.write(f"{_fmt17(prob_lo[d] + (lo[d] - dlo[d]) * dx[l][d])} {_fmt17(prob_lo[d] + (hi[d] - dlo[d] + 1) * dx[l][d])}\n")
            f.write(f"Level_{l}/Cell\n")


# ------------------------------------------------------------------------------------------------ checkpoints
@dataclass
class CheckpointHeader:
    finest_level: int
    istep: List[int]
    dt: List[float]
    tNew: List[float]
    grids: List[List[Box]]


def read_checkpoint_header(name: str) -> CheckpointHeader:
    """reference src/simulation.hpp:2676-2735"""
    with open(os.path.join(name, "Header")) as f:
        L = f.read().split("\n")
    assert L[0] == "Checkpoint file for QuokkaCode", L[0]
    finest = int(L[1])
    istep = [int(x) for x in L[2].split()]
    dt = [float(x) for x in L[3].split()]
    tnew = [float(x) for x in L[4].split()]
    p, grids = 5, []
    for _ in range(finest + 1):
        n = int(L[p].lstrip("(").split()[0])
        grids.append([_parse_box(L[p + 1 + b]) for b in range(n)])
        assert L[p + 1 + n].strip() == ")"
        p += n + 2
    return CheckpointHeader(finest, istep, dt, tnew, grids)


def write_checkpoint_header(name: str, h: CheckpointHeader, ndim: int):
    """reference src/simulation.hpp:2596-2640"""
    with open(os.path.join(name, "Header"), "w") as f:
        f.write("Checkpoint file for QuokkaCode\n")
        f.write(f"{h.finest_level}\n")
        f.write("".join(f"{s} " for s in h.istep) + "\n")
        f.write("".join(_fmt17(v) + " " for v in h.dt) + "\n")
        f.write("".join(_fmt17(v) + " " for v in h.tNew) + "\n")
        for lev in range(h.finest_level + 1):
            f.write(_boxarray_str(h.grids[lev], ndim) + "\n")
    with open(os.path.join(name, "metadata.yaml"), "w") as f:
        f.write("{}\n")


def read_checkpoint(name: str):
    h = read_checkpoint_header(name)
    return h, [read_vismf(os.path.join(name, f"Level_{l}", "Cell")) for l in range(h.finest_level + 1)]


# ------------------------------------------------------------------------------------------------ fcompare
def compare_plotfiles(a: str, b: str) -> Dict[str, float]:
    """What amrex's fcompare reports: per variable, the largest absolute difference over all levels (the grids must agree)."""
    A, B = read_plotfile(a), read_plotfile(b)
    assert A.varnames == B.varnames and A.finest_level == B.finest_level, "plotfiles are not comparable"
    out = {v: 0.0 for v in A.varnames}
    for la, lb in zip(A.levels, B.levels):
        assert la.boxes == lb.boxes, "grids differ"
        for fa, fb in zip(la.fabs, lb.fabs):
            for n, v in enumerate(A.varnames):
                out[v] = max(out[v], float(np.abs(fa[n] - fb[n]).max()))
    return out


# ------------------------------------------------------------------------------------------------ writers for the Python drivers
HYDRO_NAMES = ["gasDensity", "x-GasMomentum", "y-GasMomentum", "z-GasMomentum", "gasEnergy", "gasInternalEnergy"]


def component_names(ncomp_cc: int) -> List[str]:
    """QuokkaSimulation::defineComponentNames (reference src/QuokkaSimulation.hpp:283-310), no passive scalars"""
    names = list(HYDRO_NAMES)
    for g in range((ncomp_cc - 6) // 4):
        names += [f"radEnergy-Group{g}", f"x-RadFlux-Group{g}", f"y-RadFlux-Group{g}", f"z-RadFlux-Group{g}"]
    return names


def _levels_of(sim):
    """(level objects, per-level istep, per-level time) of a HydroSimulation or an AmrSimulation"""
    if hasattr(sim, "levels"):
        return list(sim.levels), list(sim.istep[:len(sim.levels)]), [L.t_new for L in sim.levels]
    return [sim], [sim.istep], [sim.tNew_]


def _comm_helpers(nranks: int):
    if nranks == 1:
        return None, None
    import torch
    import torch.distributed as dist

    def reduce_minmax(mins, maxs):
        a, b = torch.from_numpy(mins.copy()), torch.from_numpy(maxs.copy())
        dist.all_reduce(a, op=dist.ReduceOp.MIN)
        dist.all_reduce(b, op=dist.ReduceOp.MAX)
        return a.numpy(), b.numpy()

    return reduce_minmax, dist.barrier


def _write_level(L, prefix: str, with_ghost: bool, reduce_minmax):
    mf, nd = L.state_new_cc_, L.geom.ndim
    ng = mf.nghost if with_ghost else 0
    fabs = [(mf.fab_numpy(b) if with_ghost else mf.valid(b).cpu().numpy()) for b in range(L.lev.nboxes)]
    write_vismf(prefix, L.all_boxes, L.owner, L.rank, fabs, mf.ncomp, ng, nd, reduce_minmax)


def WritePlotFile(sim, name: str):
    """AMRSimulation::WritePlotFile (reference src/simulation.hpp:2294-2336): state_new_cc_ of every level without ghost cells"""
    levels, steps, _ = _levels_of(sim)
    g0 = levels[0].geom
    reduce_minmax, barrier = _comm_helpers(levels[0].nranks)
    _prebuild(name, len(levels), levels[0].rank, barrier)
    if levels[0].rank == 0:
        domains = [([0, 0, 0], [L.geom.n_cell[d] - 1 if d < g0.ndim else 0 for d in range(3)]) for L in levels]
        write_plotfile_header(name, component_names(levels[0].ncomp_cc), g0.ndim, float(levels[0].t_new if hasattr(levels[0], "t_new") else sim.tNew_), g0.prob_lo,
                              g0.prob_hi, domains, steps, [L.geom.dx for L in levels], [L.all_boxes for L in levels])
        with open(os.path.join(name, "metadata.yaml"), "w") as f:
            f.write("{}\n")
    for l, L in enumerate(levels):
        _write_level(L, os.path.join(name, f"Level_{l}", "Cell"), False, reduce_minmax)
    if barrier is not None:
        barrier()


def WriteCheckpointFile(sim, name: str):
    """AMRSimulation::WriteCheckpointFile (reference src/simulation.hpp:2564-2666): state_new_cc_ of every level with its ghost cells"""
    levels, steps, times = _levels_of(sim)
    nd = levels[0].geom.ndim
    reduce_minmax, barrier = _comm_helpers(levels[0].nranks)
    _prebuild(name, len(levels), levels[0].rank, barrier)
    if levels[0].rank == 0:
        nmax = (sim.max_level + 1) if hasattr(sim, "levels") else 1
        istep = list(sim.istep) if hasattr(sim, "levels") else [sim.istep]
        dt = list(sim.dt_) if hasattr(sim, "levels") else [sim.dt_]
        tnew = times + [0.0] * (nmax - len(times))
        write_checkpoint_header(name, CheckpointHeader(len(levels) - 1, istep, dt, tnew, [L.all_boxes for L in levels]), nd)
        link = os.path.join(os.path.dirname(name) or ".", "last_chk")
        if os.path.islink(link):
            os.remove(link)
        os.symlink(os.path.basename(name), link)
    for l, L in enumerate(levels):
        _write_level(L, os.path.join(name, f"Level_{l}", "Cell"), True, reduce_minmax)
    if barrier is not None:
        barrier()


def ReadCheckpointLevel0(sim, name: str):
    """Level 0 of AMRSimulation::ReadCheckpointFile for a HydroSimulation: the simulation keeps its own boxes, the data arrive as
    ParallelCopy would bring them (reference src/simulation.hpp:2795-2801); every rank reads the fabs that overlap its boxes."""
    import torch
    h = read_checkpoint_header(name)
    src = read_vismf(os.path.join(name, "Level_0", "Cell"))
    mf = sim.state_new_cc_
    for b in range(sim.lev.nboxes):
        a = mf.fab_numpy(b).copy()
        beg = mf.begins[b]
        end = [beg[d] + a.shape[3 - d] - 1 for d in range(3)]
        for use_valid in (False, True):
            for (vlo, vhi), (flo, fhi), fab in zip(src.boxes, src.fabboxes, src.fabs):
                lo0, hi0 = (vlo, vhi) if use_valid else (flo, fhi)
                lo = [max(lo0[d], beg[d]) for d in range(3)]
                hi = [min(hi0[d], end[d]) for d in range(3)]
                if any(lo[d] > hi[d] for d in range(3)):
                    continue
                dst = tuple([slice(None)] + [slice(lo[d] - beg[d], hi[d] - beg[d] + 1) for d in (2, 1, 0)])
                frm = tuple([slice(None)] + [slice(lo[d] - flo[d], hi[d] - flo[d] + 1) for d in (2, 1, 0)])
                a[dst] = fab[frm]
        mf.set_fab(b, a)
    sim.istep, sim.dt_, sim.tNew_ = h.istep[0], h.dt[0], h.tNew[0]
    sim.fillBoundaryConditions(sim.state_new_cc_)
    sim.state_old_cc_.copy_from(sim.state_new_cc_)
    return h
