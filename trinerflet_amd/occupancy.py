"""The occupancy window: the part of each plane a sample of a batch marched through the current density grid can read.

Samples only exist inside occupied grid cells (raymarching.cu:312-480: the march skips unoccupied cells), so the box of
the occupied cells of all cascades, projected on the three planes and grown by the bilinear corner, bounds every texel
the training forward reads and every texel its gradient reaches.  TrainStep rebuilds / differentiates / updates only
that window; the module path (nerf/network.py) converts only it to the sampler's layout.  No reference counterpart: the
reference rebuilds and samples whole planes (triplane_encoder.py:364-439).

    request(...)             two small kernels over the bitfield + an asynchronous 2-KB read-back behind an event
    finish(request, ...)     waits for the event; -> (window or None, row extents or None)
    window(...)              the two in one call (synchronises the host with the stream)
"""
import math

import numpy as np
import torch

from . import _lib as L


def request(bitfield, cascade, grid_size, bound, R, rows=False, host=None):
    """bitfield: uint8 [cascade * H^3 / 8].  rows: also the per-plane, per-8-texel-row-group column extents
    (tnl_occupancy_row_extents).  host: a pinned int32 buffer of the right size to reuse (a pinned allocation is a driver
    call), or None.  Returns the request tuple finish() takes."""
    dev = bitfield.device
    bits = bitfield.view(cascade, -1)                                              # [casc, H^3/8] uint8
    # one device buffer for both results, initialised by device fills, ONE read-back:
    #   [casc][6] bounding boxes {H+1, H+1, H+1, -1, -1, -1} | [3][R/8][2] row pieces {INT_MAX, -1}
    rows = bool(rows) and R % 8 == 0
    nb, ne = cascade * 6, (3 * (R // 8) * 2 if rows else 0)
    buf = torch.empty(nb + ne, dtype=torch.int32, device=dev)
    buf[:nb].view(cascade, 2, 3)[:, 0].fill_(grid_size + 1)
    buf[:nb].view(cascade, 2, 3)[:, 1].fill_(-1)
    bounds = buf[:nb].view(cascade, 6)
    L.check(L.lib().tnl_occupancy_bounds(L.ptr(bits), L.u32(bits.shape[1]), L.u32(cascade), L.ptr(bounds), L.stream()),
            "occupancy_bounds")
    if rows:
        # per plane and 8-texel row group the columns a sample can touch
        ext = buf[nb:].view(-1, 2)
        ext[:, 0].fill_(0x7fffffff)
        ext[:, 1].fill_(-1)
        L.check(L.lib().tnl_occupancy_row_extents(L.ptr(bits), L.u32(bits.shape[1]), L.u32(cascade), L.u32(grid_size),
                                                  L.f32(float(bound)), L.u32(R), L.ptr(ext), L.stream()),
                "occupancy_row_extents")
    if host is None or host.numel() != nb + ne:
        host = torch.empty(nb + ne, dtype=torch.int32, pin_memory=True)
    host.copy_(buf, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return (host, buf, ev, nb, rows)


def window_from_bounds(boxes, cascade, grid_size, bound, R):
    """boxes: [cascade][6] cell-index bounding boxes {min xyz, max xyz} (max < 0: cascade empty).  Returns the window
    [ox0, ox1, ox2, oy0, oy1, oy2, rw, rh] in texels (multiples of 64; one size for the three planes), or None when it
    covers more than 80 % of a plane."""
    vals = [float("inf")] * 3 + [float("-inf")] * 3            # world-space box over the cascades
    for k, bk in enumerate(boxes):
        if bk[3] < 0:
            continue                                            # no occupied cell in this cascade
        sk = min(2.0 ** k, float(bound))
        for a in range(3):
            vals[a] = min(vals[a], (bk[a] / grid_size * 2 - 1) * sk)
            vals[3 + a] = max(vals[3 + a], ((bk[3 + a] + 1) / grid_size * 2 - 1) * sk)
    if not all(math.isfinite(v) for v in vals):
        vals = [0.0] * 6                                                                # empty grid: no samples
    b = float(bound)

    def texels(a):   # axis a -> [first, end) texel range incl. the +1 corner and one texel of slack each side
        f0 = (min(max(vals[a] / b, -1.0), 1.0) + 1) / 2 * (R - 1)
        f1 = (min(max(vals[3 + a] / b, -1.0), 1.0) + 1) / 2 * (R - 1)
        t0 = max(int(math.floor(f0)) - 1, 0) // 64 * 64
        t1 = min((int(math.floor(f1)) + 3 + 63) // 64 * 64, R)
        return t0, t1
    xa, ya = (0, 0, 1), (2, 1, 2)   # plane p samples (axis xa[p] -> texel x, axis ya[p] -> texel y)
    xr = [texels(a) for a in xa]
    yr = [texels(a) for a in ya]
    rw = max(t1 - t0 for t0, t1 in xr)
    rh = max(t1 - t0 for t0, t1 in yr)
    if rw * rh > 0.8 * R * R:
        return None
    ox = [min(t0, R - rw) for t0, _ in xr]
    oy = [min(t0, R - rh) for t0, _ in yr]
    return ox + oy + [rw, rh]


def finish(req, cascade, grid_size, bound, R):
    """-> (window or None, row extents int64 [3][R/8][2] or None)."""
    host_t, _, ev, nb, rows = req
    ev.synchronize()
    host = host_t.numpy().copy()     # the pinned buffer may be reused by the next request
    ext = host[nb:].reshape(3, R // 8, 2).astype(np.int64) if rows else None
    return window_from_bounds(host[:nb].reshape(cascade, 6).tolist(), cascade, grid_size, bound, R), ext


def window(bitfield, cascade, grid_size, bound, R):
    return finish(request(bitfield, cascade, grid_size, bound, R), cascade, grid_size, bound, R)[0]


def level_windows(roi, J, R):
    """Per IDWT level (0 = coarsest) the window of its OUTPUT that the next level needs: the finest level's window is the
    occupancy window `roi`; below it, the next window halved and grown by 8 texels (the kernels stage a 4-texel halo),
    aligned outward to 64 with a common size over the planes.  None where the whole plane is needed anyway."""
    wins = [None] * J
    if roi is None or J == 0:
        return wins
    wins[J - 1] = list(roi)
    for lvl in range(J - 2, -1, -1):
        nxt = wins[lvl + 1]
        m = R >> (J - 1 - lvl)              # output size of this level
        if nxt is None or m % 64 != 0:
            break
        lo_x = [max((nxt[p] // 2 - 8) // 64 * 64, 0) for p in range(3)]
        lo_y = [max((nxt[3 + p] // 2 - 8) // 64 * 64, 0) for p in range(3)]
        hi_x = [min(((nxt[p] + nxt[6]) // 2 + 8 + 63) // 64 * 64, m) for p in range(3)]
        hi_y = [min(((nxt[3 + p] + nxt[7]) // 2 + 8 + 63) // 64 * 64, m) for p in range(3)]
        rw = max(h - l for l, h in zip(lo_x, hi_x))
        rh = max(h - l for l, h in zip(lo_y, hi_y))
        if rw * rh > 0.8 * m * m:
            break
        wins[lvl] = [min(l, m - rw) for l in lo_x] + [min(l, m - rh) for l in lo_y] + [rw, rh]
    return wins


def live_rects(wins, rects, sizes, col_align=32):
    """Per level the rectangle (per plane origin, common size, columns in multiples of col_align, rows of 8; the level's own
    coordinates) holding everything the windowed rebuild reads -- the level's output window `wins[lvl]` halved and grown by
    6 (the longest filter, bior6.8, reaches 4-5 coefficients to either side) -- and everything the windowed adjoint writes
    (`rects[lvl]`).  None: the whole level stays live.  sizes[lvl] = the level's n.  (TrainStep._live_rects and
    optim.FusedAdamL1's deferred pass share this rule; tests/test_adam_deferred_gpu.py poisons everything outside.)"""
    live = [None] * len(sizes)
    for lvl, n in enumerate(sizes):
        w, r = wins[lvl], rects[lvl]
        if w is None or r is None:
            continue

        def span(o, size, ro, rsize, al):
            lo = min(ro, max(o // 2 - 6, 0)) // al * al
            hi = min((max(ro + rsize, (o + size) // 2 + 6) + al - 1) // al * al, n)
            return lo, hi
        xs = [span(w[p], w[6], r[p], r[6], col_align) for p in range(3)]
        ys = [span(w[3 + p], w[7], r[3 + p], r[7], 8) for p in range(3)]
        rw = max(h - l for l, h in xs)
        rh = max(h - l for l, h in ys)
        if rw * rh > 0.8 * n * n:
            continue
        live[lvl] = [min(l, n - rw) for l, _ in xs] + [min(l, n - rh) for l, _ in ys] + [rw, rh]
    return live
