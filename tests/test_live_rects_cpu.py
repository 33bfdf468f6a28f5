"""Host logic of the live / deferred split of the optimiser pass (TrainStep._live_rects, _forward_windows): for random
occupancy windows the live rectangle of every level must contain the gradient's rectangle and everything the windowed
plane rebuild reads, keep the alignments the kernels rely on, and give the level up (None) when it is not worth it."""
import types

import numpy as np
import pytest
import torch

from trinerflet_amd.train import TrainStep


def _dummy(R, J, roi):
    ns = types.SimpleNamespace()
    ns.R, ns.J, ns._roi = R, J, roi
    ns.coef = types.SimpleNamespace(params=[torch.empty(1, 1, 1, R >> (J - lvl), R >> (J - lvl)) for lvl in range(J)])
    ns._forward_windows = lambda: TrainStep._forward_windows(ns)
    return ns


@pytest.mark.parametrize("R,J", [(2048, 5), (1024, 4), (512, 3)])
def test_live_rectangles_cover_window_footprint_and_gradient_rectangle(R, J):
    rng = np.random.default_rng(R)
    seen_live = 0
    for trial in range(200):
        rw, rh = (int(rng.integers(1, R // 64)) * 64 for _ in range(2))
        ox = [int(rng.integers(0, (R - rw) // 64 + 1)) * 64 for _ in range(3)]
        oy = [int(rng.integers(0, (R - rh) // 64 + 1)) * 64 for _ in range(3)]
        roi = ox + oy + [rw, rh]
        ts = _dummy(R, J, roi)
        wins = ts._forward_windows()
        assert wins[J - 1] == roi
        # a plausible chain of gradient rectangles: each level's window halved, grown by the filter reach, 8-aligned
        rects, win = [None] * J, roi
        for lvl in reversed(range(J)):
            n = R >> (J - lvl)
            lo_x = [max((win[p] - 16) // 2 // 8 * 8, 0) for p in range(3)]
            lo_y = [max((win[3 + p] - 16) // 2 // 8 * 8, 0) for p in range(3)]
            w = min(((win[6] + 32) // 2 + 15) // 8 * 8, n)
            h = min(((win[7] + 32) // 2 + 15) // 8 * 8, n)
            rects[lvl] = [min(l, n - w) for l in lo_x] + [min(l, n - h) for l in lo_y] + [w, h]
            win = rects[lvl]
        live = TrainStep._live_rects(ts, rects)
        for lvl in range(J):
            n = R >> (J - lvl)
            lv, r, w = live[lvl], rects[lvl], wins[lvl]
            if lv is None:
                continue
            seen_live += 1
            assert w is not None                                   # a level rebuilt whole is never deferred
            assert lv[6] % 32 == 0 and lv[7] % 8 == 0 and lv[6] * lv[7] <= 0.8 * n * n
            for p in range(3):
                assert lv[p] % 32 == 0 and lv[3 + p] % 8 == 0 and 0 <= lv[p] and lv[p] + lv[6] <= n and lv[3 + p] + lv[7] <= n
                # the gradient's rectangle
                assert lv[p] <= r[p] and r[p] + r[6] <= lv[p] + lv[6]
                assert lv[3 + p] <= r[3 + p] and r[3 + p] + r[7] <= lv[3 + p] + lv[7]
                # what the windowed rebuild reads: the level's output window halved, +-5 coefficients (bior6.8)
                assert lv[p] <= max(w[p] // 2 - 5, 0) and min((w[p] + w[6]) // 2 + 5, n) <= lv[p] + lv[6]
                assert lv[3 + p] <= max(w[3 + p] // 2 - 5, 0) and min((w[3 + p] + w[7]) // 2 + 5, n) <= lv[3 + p] + lv[7]
    assert seen_live > 50
