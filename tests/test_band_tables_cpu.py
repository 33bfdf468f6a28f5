"""Host logic of the occupancy pieces (TrainStep._level_needs, _band_tables, _forward_spans): from per-row-group
column pieces of the plane grid, level by level, to the band tables of the optimiser pass and the span tables of the
IDWT kernels.  Checked against the plain definition on boolean masks: a coefficient is needed when a needed position of
the next finer grid lies within the filter reach of its double."""
import types

import numpy as np
import pytest
import torch

from trinerflet_amd.train import TrainStep

BIG = 0x7fffffff


def _dummy(R, J, ext):
    ns = types.SimpleNamespace()
    ns.R, ns.J, ns.base_res, ns.live_bands, ns.live_col_align = R, J, 0, True, 32
    ns.dev = torch.device("cpu")
    ns.coef = types.SimpleNamespace(params=[torch.empty(1, 1, 1, R >> (J - lvl), R >> (J - lvl)) for lvl in range(J)])
    ns._band_cache, ns._row_ext = {}, ext
    ns._level_needs = lambda: TrainStep._level_needs(ns)
    return ns


def _mask_of(table, n):
    """[3, n, n] bool from a [3, n/8, 2] table of column pieces per 8 rows."""
    m = np.zeros((3, n, n), bool)
    for p in range(3):
        for g in range(n // 8):
            lo, hi = int(table[p, g, 0]), int(table[p, g, 1])
            if hi > lo:
                m[p, 8 * g:8 * g + 8, lo:hi] = True
    return m


def _blob_ext(R, rng):
    """Row pieces of three random convex blobs (ellipses), one per plane, as tnl_occupancy_row_extents would report."""
    ext = np.empty((3, R // 8, 2), np.int64)
    ext[..., 0], ext[..., 1] = BIG, -1
    for p in range(3):
        cx, cy = rng.integers(R // 4, 3 * R // 4, 2)
        ax, ay = rng.integers(R // 16, R // 4, 2)
        for g in range(R // 8):
            ys = np.arange(8 * g, 8 * g + 8)
            t = 1 - ((ys - cy) / ay) ** 2
            if (t > 0).any():
                half = int(ax * np.sqrt(t.max()))
                ext[p, g] = max(cx - half - 1, 0), min(cx + half + 3, R)
    return ext


@pytest.mark.parametrize("R,J", [(512, 3), (1024, 4)])
def test_level_needs_contain_the_definition(R, J):
    rng = np.random.default_rng(R + J)
    for trial in range(5):
        ext = _blob_ext(R, rng)
        ts = _dummy(R, J, ext)
        needs = ts._level_needs()
        assert needs is not None and len(needs) == J
        finer = _mask_of(ext, R)
        for lvl in reversed(range(J)):
            n = R >> (J - lvl)
            got = _mask_of(needs[lvl], n)
            # definition: (r, c) needed iff a needed (r', c') of the finer grid has |r - r'/2| <= 6 and |c - c'/2| <= 6
            want = np.zeros((3, n, n), bool)
            half = finer[:, 0::2, :] | finer[:, 1::2, :]
            half = half[:, :, 0::2] | half[:, :, 1::2]
            for p in range(3):
                rr, cc = np.nonzero(half[p])
                if rr.size:
                    box = np.zeros((n, n), bool)
                    r0s, r1s = np.maximum(rr - 6, 0), np.minimum(rr + 7, n)
                    c0s, c1s = np.maximum(cc - 6, 0), np.minimum(cc + 7, n)
                    # (sample a subset of the positions: the property is monotone, containment of each box suffices)
                    for k in rng.choice(rr.size, size=min(rr.size, 400), replace=False):
                        box[r0s[k]:r1s[k], c0s[k]:c1s[k]] = True
                    want[p] = box
            assert not (want & ~got).any(), (lvl, int((want & ~got).sum()))
            # ... and not absurdly more than its own row-wise hull grown by one row group and the reach
            assert got.sum() <= 3.0 * max(want.sum(), 1) + 3 * 64 * n
            finer = got


@pytest.mark.parametrize("R,J", [(512, 3), (2048, 5)])
def test_band_tables_cover_the_needs_inside_the_live_rectangle(R, J):
    rng = np.random.default_rng(7 * R)
    made = 0
    for trial in range(12):
        ext = _blob_ext(R, rng)
        ts = _dummy(R, J, ext)
        needs = ts._level_needs()
        live = [None] * J
        for lvl in range(J):
            n = R >> (J - lvl)
            if n < 64:
                continue
            # a live rectangle around the needs of all three planes (common size, 32 / 8 aligned), as _live_rects makes
            boxes = []
            for p in range(3):
                g = np.nonzero(needs[lvl][p, :, 1] > needs[lvl][p, :, 0])[0]
                lo, hi = needs[lvl][p, g, 0].min(), needs[lvl][p, g, 1].max()
                boxes.append((lo // 32 * 32, min((hi + 31) // 32 * 32, n), 8 * g.min(), 8 * g.max() + 8))
            rw = max(b[1] - b[0] for b in boxes)
            rh = max(b[3] - b[2] for b in boxes)
            live[lvl] = [int(min(b[0], n - rw)) for b in boxes] + [int(min(b[2], n - rh)) for b in boxes] + [int(rw), int(rh)]
        tabs = TrainStep._band_tables(ts, live)
        assert TrainStep._band_tables(ts, live) is tabs                    # cached per set of rectangles
        for lvl, (lv, bt) in enumerate(zip(live, tabs)):
            if bt is None:
                continue
            made += 1
            n = R >> (J - lvl)
            dev_tbl, quads, tbl, spans = bt
            nb = lv[7] // 8
            assert tbl.size == 5 * nb + 1 and torch.equal(dev_tbl, torch.from_numpy(tbl)) and int(tbl[nb]) == quads
            assert quads <= 0.92 * lv[7] * lv[6] // 4 and (np.diff(tbl[:nb + 1]) == 8 * tbl[nb + 1:2 * nb + 1]).all()
            need = _mask_of(needs[lvl], n)
            sp = spans.numpy().reshape(3, n // 8, 2)
            for p in range(3):
                piece = np.zeros((n, n), bool)
                for b in range(nb):
                    w, x0 = 4 * int(tbl[nb + 1 + b]), int(tbl[2 * nb + 1 + p * nb + b])
                    assert x0 % 32 == 0 and w % 32 == 0 and lv[p] <= x0 and x0 + w <= lv[p] + lv[6]
                    piece[lv[3 + p] + 8 * b:lv[3 + p] + 8 * b + 8, x0:x0 + w] = True
                    g = lv[3 + p] // 8 + b                                 # the adjoint's span table states the same pieces
                    assert (w == 0 and sp[p, g, 1] <= sp[p, g, 0]) or (sp[p, g, 0] == x0 and sp[p, g, 1] == x0 + w)
                assert not (need[p] & ~piece).any()                        # every needed coefficient is live
                outside = np.ones(n // 8, bool)
                outside[lv[3 + p] // 8:lv[3 + p] // 8 + nb] = False
                assert (sp[p, outside, 1] <= sp[p, outside, 0]).all()
    assert made >= 6


def test_forward_spans_halve_the_next_grid():
    R, J = 1024, 4
    ext = _blob_ext(R, np.random.default_rng(3))
    ts = _dummy(R, J, ext)
    fw, plane = TrainStep._forward_spans(ts)
    assert torch.equal(plane, torch.from_numpy(ext.astype(np.int32).reshape(-1)))
    needs = ts._level_needs()
    for lvl in range(J):
        n = R >> (J - lvl)
        got = _mask_of(fw[lvl].numpy().reshape(3, n // 8, 2), n)
        out = _mask_of(ext if lvl == J - 1 else needs[lvl + 1], 2 * n)     # what the level's output must hold
        produced = np.repeat(np.repeat(got, 2, axis=1), 2, axis=2)         # a coarse position yields its four outputs
        assert not (out & ~produced).any()
