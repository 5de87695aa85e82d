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
