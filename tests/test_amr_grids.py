"""Host logic of the AMR grid generation (quokka_amd/amr_simulation.py), no GPU: buffered tags -> blocking-factor tiles -> boxes."""
import numpy as np
import pytest

from quokka_amd.amr_simulation import boxes_from_tags, boxes_from_tiles, covered_mask, dilate


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_boxes_cover_buffered_tags_and_respect_limits(seed):
    rng = np.random.default_rng(seed)
    n, bf, mgs, nbuf = 64, 16, 32, 3
    tags = rng.random((n, n, n)) < 2e-4
    tags[10:14, 30:33, 50:52] = True
    boxes = boxes_from_tags(tags, 3, nbuf, bf, mgs)
    fine = covered_mask(boxes, (2 * n,) * 3)
    count = np.zeros((2 * n,) * 3, dtype=np.int32)
    for lo, hi in boxes:
        assert all(lo[d] % bf == 0 and (hi[d] + 1) % bf == 0 and hi[d] - lo[d] + 1 <= mgs and 0 <= lo[d] and hi[d] < 2 * n for d in range(3))
        count[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] += 1
    assert count.max() == 1, "boxes overlap"
    refined = fine.reshape(n, 2, n, 2, n, 2).all(axis=(1, 3, 5))
    need = dilate(tags, nbuf, 3)
    assert not (need & ~refined).any(), "a buffered tag is not refined"
    # no tile without a buffered tag is refined (the clustering adds nothing but the tile granularity)
    tile = bf // 2
    need_t = need.reshape(n // tile, tile, n // tile, tile, n // tile, tile).any(axis=(1, 3, 5))
    ref_t = refined.reshape(n // tile, tile, n // tile, tile, n // tile, tile).all(axis=(1, 3, 5))
    assert np.array_equal(need_t, ref_t)


def test_dilate_is_max_norm_ball():
    m = np.zeros((9, 9, 9), dtype=bool)
    m[4, 4, 4] = True
    d = dilate(m, 2, 3)
    assert d.sum() == 125 and d[2:7, 2:7, 2:7].all()
    assert dilate(m, 2, 1).sum() == 5 and dilate(m, 0, 3).sum() == 1


def test_allowed_mask_and_tiles():
    t = np.zeros((4, 4, 4), dtype=bool)
    t[0, 0, :] = True
    boxes = boxes_from_tiles(t, 3, 8, 16)
    assert boxes == [([0, 0, 0], [15, 7, 7]), ([16, 0, 0], [31, 7, 7])]
    tags = np.zeros((16, 16, 16), dtype=bool)
    tags[1, 1, 1] = True
    allowed = np.zeros_like(tags)
    assert boxes_from_tags(tags, 3, 0, 8, 16, allowed=allowed) == []


def test_interleaved_level0_map_spreads_every_neighbourhood():
    """level-0 box -> rank map of the multi-rank AMR driver: any 2x2x2 block of neighbouring level-0 boxes lies on min(8, nranks)
    different ranks (refined boxes stay with their level-0 ancestor, so a localised refined region is shared by all ranks), and
    the boxes are split evenly"""
    from quokka_amd.simulation import chop_domain, distribute_boxes_interleaved
    n_cell, mgs = [256, 256, 256], [64, 64, 64]
    boxes = chop_domain(n_cell, mgs)
    for nranks in (2, 4, 8):
        owner = distribute_boxes_interleaved(boxes, nranks, n_cell, mgs)
        assert sorted(set(owner)) == list(range(nranks)) and max(owner.count(r) for r in range(nranks)) == len(boxes) // nranks
        at = {tuple(lo[d] // 64 for d in range(3)): o for (lo, hi), o in zip(boxes, owner)}
        for k in range(3):
            for j in range(3):
                for i in range(3):
                    block = {at[(i + a, j + b, k + c)] for a in (0, 1) for b in (0, 1) for c in (0, 1)}
                    assert len(block) == min(8, nranks)


# ------------------------------------------------------------------------------------------------ Berger-Rigoutsos clustering
def _check_boxes(t, boxes, bf, mgs):
    """boxes are disjoint, blocking-factor aligned, <= max_grid_size, and cover every flagged tile; returns the efficiency"""
    tz, ty, tx = t.shape
    cover = np.zeros_like(t, dtype=np.int32)
    for lo, hi in boxes:
        assert all(lo[d] % bf == 0 and (hi[d] + 1) % bf == 0 and hi[d] - lo[d] + 1 <= mgs for d in range(3)), (lo, hi)
        cover[lo[2] // bf:hi[2] // bf + 1, lo[1] // bf:hi[1] // bf + 1, lo[0] // bf:hi[0] // bf + 1] += 1
    assert cover.max() <= 1, "boxes overlap"
    assert not (t & (cover == 0)).any(), "a flagged tile is not covered"
    return t.sum() / max(cover.sum(), 1)


def test_berger_rigoutsos_hole_cut_separates_two_blobs():
    """two blobs separated by empty planes: one hole cut, two boxes, efficiency 1"""
    t = np.zeros((8, 8, 8), dtype=bool)
    t[0:2, 0:2, 0:3] = True
    t[5:8, 4:8, 6:8] = True
    boxes = boxes_from_tiles(t, 3, 8, 64, grid_eff=0.7)
    assert sorted(boxes) == [([0, 0, 0], [23, 15, 15]), ([48, 32, 40], [63, 63, 63])]
    assert _check_boxes(t, boxes, 8, 64) == 1.0


def test_berger_rigoutsos_l_shape_meets_the_efficiency():
    """an L of tiles: the bounding box has efficiency 7/16 < 0.7; the inflection / bisection cuts must bring every box above grid_eff"""
    t = np.zeros((1, 4, 4), dtype=bool)
    t[0, 0, :] = True
    t[0, :, 0] = True
    boxes = boxes_from_tiles(t, 3, 8, 64, grid_eff=0.7)
    eff = _check_boxes(t, boxes, 8, 64)
    assert eff >= 0.7, (eff, boxes)
    # a full block stays one box, cut only by max_grid_size (ceil(len / max) nearly equal pieces)
    full = np.ones((2, 2, 5), dtype=bool)
    boxes = boxes_from_tiles(full, 3, 8, 16, grid_eff=0.7)
    assert len(boxes) == 3 and sorted(hi[0] - lo[0] + 1 for lo, hi in boxes) == [8, 16, 16]
    assert _check_boxes(full, boxes, 8, 16) == 1.0


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_berger_rigoutsos_random_shells(seed):
    """a thick spherical shell of flagged tiles (what a blast wave tags): every box meets grid_eff (single tiles are exempt by
    construction), nothing flagged is lost, and the rule refines fewer tiles than the bounding box and no more than ~1/grid_eff of them"""
    rng = np.random.default_rng(seed)
    n = 16
    k, j, i = np.meshgrid(*(np.arange(n),) * 3, indexing="ij")
    r = np.sqrt((i + 0.5) ** 2 + (j + 0.5) ** 2 + (k + 0.5) ** 2)
    R = 6.0 + 6.0 * rng.random()
    t = (np.abs(r - R) < 1.2)
    boxes = boxes_from_tiles(t, 3, 32, 128, grid_eff=0.7)
    eff = _check_boxes(t, boxes, 32, 128)
    assert eff >= 0.7 * 0.95, eff  # (max_grid_size cuts do not change the covered set; simplify never lowers the efficiency)
    greedy = boxes_from_tiles(t, 3, 32, 128)
    assert _check_boxes(t, greedy, 32, 128) == 1.0  # the round-1 rule refines exactly the flagged tiles, in many more boxes
    assert len(boxes) <= len(greedy)


def test_berger_rigoutsos_respects_the_nesting_domain():
    """a cluster box may hold unflagged tiles; where such a tile lies outside the allowed (proper-nesting) region the box is bisected until
    every piece is allowed, and nothing flagged is lost"""
    t = np.zeros((1, 4, 4), dtype=bool)
    t[0, 0:3, 0:4] = True
    t[0, 1, 1] = False  # 11 of 12: efficiency 0.92, one box with a hole ...
    allowed = np.ones_like(t)
    allowed[0, 1, 1] = False  # ... which is forbidden
    free = boxes_from_tiles(t, 3, 8, 64, grid_eff=0.7)
    assert free == [([0, 0, 0], [31, 23, 7])]
    boxes = boxes_from_tiles(t, 3, 8, 64, grid_eff=0.7, allowed=allowed)
    assert _check_boxes(t, boxes, 8, 64) == 1.0
    cover = np.zeros_like(t)
    for lo, hi in boxes:
        cover[lo[2] // 8:hi[2] // 8 + 1, lo[1] // 8:hi[1] // 8 + 1, lo[0] // 8:hi[0] // 8 + 1] = True
    assert not cover[0, 1, 1]


# ------------------------------------------------------------------------------------------------ boxes -> ranks (levels with their own distribution)
def test_chop_grids_gives_every_rank_a_box_of_the_config5_levels():
    """AmrMesh::ChopGrids as restated in amr_simulation.chop_grids: the one 64^3 box that levels 1 and 2 of tests/blast_amr_maxlev2.in are for
    the first ~55 coarse steps (max_grid_size 128, blocking_factor 32) becomes 8 boxes of 32^3 on 8 ranks, 2 / 4 boxes on 2 / 4 ranks; boxes stay
    disjoint, blocking-factor aligned and cover the same cells; a level that already has enough boxes, or one rank, is left alone."""
    from quokka_amd.amr_simulation import chop_grids
    one = [([0, 0, 0], [63, 63, 63])]
    for target, want in ((1, 1), (2, 2), (4, 4), (8, 8), (16, 8)):  # (32^3 cannot be halved again with blocking_factor 32)
        got = chop_grids(one, target, 128, 32, [512] * 3)
        assert len(got) == want, (target, got)
        cover = np.zeros((64, 64, 64), dtype=np.int32)
        for lo, hi in got:
            assert all(lo[d] % 32 == 0 and (hi[d] + 1) % 32 == 0 for d in range(3))
            cover[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] += 1
        assert (cover == 1).all()
    many = [([32 * i, 0, 0], [32 * i + 31, 31, 31]) for i in range(8)]
    assert chop_grids(many, 8, 128, 32, [512] * 3) == many
    # an L of two boxes, 4 ranks: both boxes are cut, the longest direction first
    got = chop_grids([([0, 0, 0], [63, 31, 31]), ([0, 32, 0], [31, 63, 31])], 4, 64, 16, [128] * 3)
    assert len(got) >= 4 and sum(np.prod([hi[d] - lo[d] + 1 for d in range(3)]) for lo, hi in got) == 64 * 32 * 32 + 32 * 32 * 32


def test_sfc_distribution_balances_cells_and_fills_the_least_loaded_ranks():
    from quokka_amd.amr_simulation import chop_grids, distribute_sfc
    boxes = chop_grids([([0, 0, 0], [63, 63, 63])], 8, 128, 32, [512] * 3)
    own = distribute_sfc(boxes, 8, [128 ** 3] * 8, unit=32)
    assert sorted(own) == list(range(8))  # one box per rank
    # fewer boxes than ranks: they go to the ranks that hold the least so far
    own = distribute_sfc(boxes[:3], 8, [5, 9, 1, 9, 9, 0, 9, 9], unit=32)
    assert sorted(own) == [0, 2, 5]
    # many equal boxes: contiguous runs along the curve, equal counts
    grid = [([16 * i, 16 * j, 16 * k], [16 * i + 15, 16 * j + 15, 16 * k + 15]) for k in range(4) for j in range(4) for i in range(4)]
    own = distribute_sfc(grid, 8, None, unit=16)
    assert [own.count(r) for r in range(8)] == [8] * 8
    for r in range(8):  # a run of the Morton curve through a 4^3 lattice of boxes = one 2x2x2 block
        mine = [grid[b][0] for b in range(64) if own[b] == r]
        assert all(max(m[d] for m in mine) - min(m[d] for m in mine) == 16 for d in range(3))
    assert distribute_sfc(grid, 1) == [0] * 64


_LAYOUT_DRIVER = r"""
#include <cstdio>
#include <iostream>
#include "qk_grid_layout.hpp"
struct B { int lo[3], hi[3]; };
int main() {  // stdin: what (chop|sfc), the parameters, the boxes; stdout: boxes or owners
    std::string what;
    while (std::cin >> what) {
        int n = 0;
        if (what == "chop") {
            int target, mgs, bf, ndim; std::array<int, 3> dom{};
            std::cin >> target >> mgs >> bf >> ndim >> dom[0] >> dom[1] >> dom[2] >> n;
            std::vector<B> boxes(n);
            for (auto &b : boxes) std::cin >> b.lo[0] >> b.lo[1] >> b.lo[2] >> b.hi[0] >> b.hi[1] >> b.hi[2];
            auto out = qkhost::chopGrids(boxes, target, mgs, bf, dom, ndim);
            std::printf("%zu", out.size());
            for (auto const &b : out) std::printf(" %d %d %d %d %d %d", b.lo[0], b.lo[1], b.lo[2], b.hi[0], b.hi[1], b.hi[2]);
            std::printf("\n");
        } else {
            int nranks, unit, nload;
            std::cin >> nranks >> unit >> nload;
            std::vector<long long> load(nload);
            for (auto &l : load) std::cin >> l;
            std::cin >> n;
            std::vector<B> boxes(n);
            for (auto &b : boxes) std::cin >> b.lo[0] >> b.lo[1] >> b.lo[2] >> b.hi[0] >> b.hi[1] >> b.hi[2];
            auto own = qkhost::distributeSfc(boxes, nranks, load, unit);
            std::printf("%zu", own.size());
            for (int o : own) std::printf(" %d", o);
            std::printf("\n");
        }
    }
    return 0;
}
"""


def test_cxx_host_grid_layout_equals_the_python_host(tmp_path):
    """quokka_amd/host/qk_grid_layout.hpp (maxSize / chopGrids / distributeSfc of the C++17 host's distributed levels) against
    amr_simulation.chop_grids / distribute_sfc on the same inputs: the fixed cases above plus seeded random box lists, every output equal."""
    import os
    import subprocess
    from quokka_amd.amr_simulation import chop_grids, distribute_sfc
    src = tmp_path / "layout.cpp"
    src.write_text(_LAYOUT_DRIVER)
    exe = tmp_path / "layout"
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "quokka_amd", "host")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", host, str(src), "-o", str(exe)], check=True)
    rng = np.random.default_rng(7)

    def fmt(boxes):
        return f"{len(boxes)} " + " ".join(" ".join(map(str, list(lo) + list(hi))) for lo, hi in boxes)

    cases, want = [], []

    def chop(boxes, target, mgs, bf, dom, ndim=3):
        cases.append(f"chop {target} {mgs} {bf} {ndim} {dom[0]} {dom[1]} {dom[2]} {fmt(boxes)}")
        out = chop_grids(boxes, target, mgs, bf, dom, ndim)
        want.append([len(out)] + [x for lo, hi in out for x in list(lo) + list(hi)])
        return out

    def sfc(boxes, nranks, load, unit):
        cases.append(f"sfc {nranks} {unit} {len(load or [])} {' '.join(map(str, load or []))} {fmt(boxes)}")
        out = distribute_sfc(boxes, nranks, load, unit=unit)
        want.append([len(out)] + out)

    one = [([0, 0, 0], [63, 63, 63])]
    for target in (1, 2, 4, 8, 16):
        sfc(chop(one, target, 128, 32, [512] * 3), 8, [128 ** 3] * 8, 32)
    chop([([0, 0, 0], [63, 31, 31]), ([0, 32, 0], [31, 63, 31])], 4, 64, 16, [128] * 3)
    chop([([0, 0, 0], [47, 31, 0])], 6, 64, 8, [64, 64, 1], ndim=2)
    grid = [([16 * i, 16 * j, 16 * k], [16 * i + 15, 16 * j + 15, 16 * k + 15]) for k in range(4) for j in range(4) for i in range(4)]
    sfc(grid, 8, None, 16)
    sfc(grid[:3], 8, [5, 9, 1, 9, 9, 0, 9, 9], 16)
    for _ in range(40):  # disjoint boxes of a random lattice with random extents (blocking factor 8), random loads
        bf = 8
        cells = [(i, j, k) for k in range(4) for j in range(4) for i in range(4)]
        pick = rng.choice(len(cells), size=int(rng.integers(1, 12)), replace=False)
        boxes = []
        for c in pick:
            lo = [32 * x for x in cells[c]]
            ext = [bf * int(rng.integers(1, 5)) for _ in range(3)]
            boxes.append((lo, [lo[d] + ext[d] - 1 for d in range(3)]))
        nranks = int(rng.integers(2, 9))
        out = chop(boxes, int(rng.integers(1, 17)), int(rng.choice([16, 32, 64])), bf, [128] * 3)
        sfc(out, nranks, [int(x) for x in rng.integers(0, 5, size=nranks) * 4096] if rng.random() < 0.7 else None, bf)
    res = subprocess.run([str(exe)], input="\n".join(cases) + "\n", capture_output=True, text=True, check=True)
    got = [[int(x) for x in ln.split()] for ln in res.stdout.strip().split("\n")]
    assert len(got) == len(want)
    for c, g, w in zip(cases, got, want):
        assert g == w, (c, g, w)


def test_register_cells_of_another_rank_sit_beside_the_face_their_coarse_flux_lives_on():
    """qk_fluxreg_create with reg_nghost = 1 (several ranks, a refined box on the rank of its level-0 ancestor): a register cell owned by another rank is
    kept in a ghost cell of a LOCAL coarse box, and CrseAdd reads the coarse flux on the face between that cell and the fine box from the same box's
    flux array — so the box must be the one whose valid region the cell adjoins in the direction of the face, never one that merely has the cell in a
    corner of its ghost ring.  The layout RadBeam reaches on four ranks after its second regrid (2-D; level 1 = four 8^2 boxes on this rank, level 2
    covering all of them): until round 6 the cells (16, 8) and (8, 16) — at the seam of two local boxes — went to the box below / left of the seam and
    CrseAdd read one row beyond the end of its flux array (profiles/round6/dist1_cxx_distributed_levels.txt)."""
    from quokka_amd.amr import FluxRegister
    from quokka_amd.multifab import Level, PlanningContext
    from quokka_amd.simulation import Geometry
    ctx = PlanningContext()
    crse_boxes = [([0, 0, 0], [7, 7, 0]), ([8, 0, 0], [15, 7, 0]), ([0, 8, 0], [7, 15, 0]), ([8, 8, 0], [15, 15, 0])]
    fine_boxes = [([8 * i, 8 * j, 0], [8 * i + 7, 8 * j + 7, 0]) for j in range(4) for i in range(4)]
    crse, fine = Level(ctx, 2, crse_boxes), Level(ctx, 2, fine_boxes)
    geom = Geometry(2, [256, 256, 1], [0, 0, 0], [2, 2, 2], [0, 0, 0])
    fr = FluxRegister(crse, fine, geom, 4, all_fine_boxes=fine_boxes, reg_nghost=1)
    items = fr.items()
    cells = set()
    for d, side, fb, cb, lo, hi, sh in items:
        blo, bhi = crse_boxes[cb]
        for e in range(2):  # inside the box in every direction but d; in d: the valid box or the one ghost cell beyond it
            g = 1 if e == d else 0
            assert blo[e] - g <= lo[e] and hi[e] <= bhi[e] + g, (d, side, fb, cb, lo, hi)
        for j in range(lo[1], hi[1] + 1):
            for i in range(lo[0], hi[0] + 1):
                assert (d, i, j) not in cells
                cells.add((d, i, j))
    assert cells == {(0, 16, j) for j in range(16)} | {(1, i, 16) for i in range(16)}  # every register cell of the fine region's two open sides, once
