"""Pieces ("spans") of the occupancy window: what the windowed IDWT levels, their adjoints and the layout change may
skip when they are told, per 8 rows, which columns anything reads (tnl_occupancy_row_extents ->
TrainStep._forward_spans / _band_tables).  Inside the pieces the results are the unrestricted calls' bits; a training
run with the pieces equals one without, bit for bit, when the plane-gradient reduction is ordered."""
import copy
import ctypes as C_

import numpy as np
import pytest
import torch

from trinerflet_amd import synthetic

pytestmark = pytest.mark.gpu
BIG = 0x7fffffff


def _disc_spans(n, centres, radius, rng):
    """[3, n/8, 2] int32: per plane a disc of the n x n grid as column pieces per 8 rows (ragged edges, an empty group)."""
    sp = np.empty((3, n // 8, 2), np.int32)
    sp[..., 0], sp[..., 1] = BIG, -1
    for p, (cx, cy) in enumerate(centres):
        for g in range(n // 8):
            dy = min(abs(8 * g - cy), abs(8 * g + 7 - cy)) if not (8 * g <= cy <= 8 * g + 7) else 0
            if dy < radius:
                half = int(np.sqrt(radius * radius - dy * dy)) + int(rng.integers(0, 5))
                sp[p, g] = max(cx - half, 0), min(cx + half + 1, n)
        sp[p, (cy // 8 + 2) % (n // 8)] = BIG, -1                     # a hole: an empty row group inside the disc
    return sp


def _piece_mask(sp, n, scale, dev):
    """[3, scale*n, scale*n] bool of the pieces (scale 2: on the level's fine side)."""
    m = torch.zeros(3, scale * n, scale * n, dtype=torch.bool, device=dev)
    for p in range(3):
        for g in range(n // 8):
            lo, hi = int(sp[p, g, 0]), int(sp[p, g, 1])
            if hi > lo:
                m[p, scale * 8 * g:scale * (8 * g + 8), scale * lo:scale * hi] = True
    return m


@pytest.mark.parametrize("half_out", [1, 0])
def test_forward_level_with_spans_equals_the_windowed_call_inside_the_pieces(cuda, half_out):
    import trinerflet_amd._lib as L
    lib = L.lib()
    n, spp, S, wave = 512, 2, 6, 4                                     # bior6.8; the column-walk kernels (n >= 512)
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(S, n, n, generator=g).to(cuda)
    yh = torch.randn(S, 3, n, n, generator=g).to(cuda) * 0.3
    win = [192, 256, 320, 256, 128, 384, 512, 448]
    rng = np.random.default_rng(5)
    centres = [((win[p] + win[6] // 2) // 2, (win[3 + p] + win[7] // 2) // 2) for p in range(3)]
    sp = _disc_spans(n, centres, 90, rng)
    spd = torch.from_numpy(sp).to(cuda)
    roi = L.roi_array(win + [spp, 0])
    res = []
    for spans in (None, spd):
        if half_out:
            out = torch.full((S, win[7], win[6]), 7.0, dtype=torch.float16, device=cuda)
        else:
            out = torch.full((S, 2 * n, 2 * n), 7.0, dtype=torch.float32, device=cuda)
        L.check(lib.tnl_idwt_level_forward_spans(L.ptr(x), L.ptr(yh), L.u32(S), L.u32(n), L.i32(wave), L.ptr(out),
                                                 L.i32(half_out), roi, L.i32(0 if half_out else 1), L.ptr(spans),
                                                 L.stream()), "forward_spans")
        res.append(out)
    mask = _piece_mask(sp, n, 2, cuda)
    kept = 0
    for s in range(S):
        p = s // spp
        ys, xs = slice(win[3 + p], win[3 + p] + win[7]), slice(win[p], win[p] + win[6])
        a, b = (res[0][s], res[1][s]) if half_out else (res[0][s][ys, xs], res[1][s][ys, xs])
        mk = mask[p][ys, xs]
        assert int(mk.sum()) > 10000
        assert torch.equal(a[mk], b[mk])
        same = (a == b) | (b == 7.0)
        assert bool(same.all())                                       # produced like the windowed call, or left alone
        kept += int(((b == 7.0) & (a != 7.0)).sum())
    assert kept > 0.15 * S * win[6] * win[7]                          # and a good part of the window was skipped


def test_adjoint_level_with_spans(cuda):
    """Band gradients inside the pieces and the LL gradient wherever it was computed: the windowed call's bits; the LL
    gradient elsewhere inside the rectangle: zero; nothing outside the rectangle is touched."""
    import trinerflet_amd._lib as L
    lib = L.lib()
    n, spp, S, wave = 512, 2, 6, 4
    g = torch.Generator(device="cpu").manual_seed(2)
    win = [192, 256, 320, 256, 128, 384, 512, 448]
    dout = torch.randn(S, win[7], win[6], generator=g).to(cuda)
    rng = np.random.default_rng(6)
    centres = [((win[p] + win[6] // 2) // 2, (win[3 + p] + win[7] // 2) // 2) for p in range(3)]
    sp = _disc_spans(n, centres, 100, rng)
    spd = torch.from_numpy(sp).to(cuda)
    res = []
    for spans in (None, spd):
        dx = torch.full((S, n, n), 7.0, device=cuda)
        dyh = torch.full((S, 3, n, n), 7.0, device=cuda)
        rect = (C_.c_int32 * 8)()
        L.check(lib.tnl_idwt_level_backward_spans(L.ptr(dout), L.u32(S), L.u32(n), L.i32(wave), L.ptr(dx), L.ptr(dyh),
                                                  L.roi_array(win + [spp, 0]), L.i32(0), rect, L.ptr(spans), L.stream()),
                "backward_spans")
        res.append((dx, dyh, list(rect)))
    assert res[0][2] == res[1][2]
    r = res[0][2]
    mask = _piece_mask(sp, n, 1, cuda)
    skipped = 0
    for s in range(S):
        p = s // spp
        inside = torch.zeros(n, n, dtype=torch.bool, device=cuda)
        inside[r[3 + p]:r[3 + p] + r[7], r[p]:r[p] + r[6]] = True
        mk = mask[p] & inside
        assert int(mk.sum()) > 5000
        for b in range(3):
            a_, b_ = res[0][1][s, b], res[1][1][s, b]
            assert torch.equal(a_[mk], b_[mk])
            assert bool(((a_ == b_) | (b_ == 7.0))[inside].all()) and bool((b_[~inside] == 7.0).all())
            skipped += int((b_[inside] == 7.0).sum())
        a_, b_ = res[0][0][s], res[1][0][s]
        assert torch.equal(a_[mk], b_[mk])
        assert bool(((a_ == b_) | (b_ == 0.0))[inside].all()) and bool((b_[~inside] == 7.0).all())
        assert bool((b_[inside] != 7.0).all())                        # every LL gradient of the rectangle was written
    assert skipped > 0.15 * S * 3 * r[6] * r[7]


def test_layout_change_with_spans(cuda):
    import trinerflet_amd._lib as L
    lib = L.lib()
    C, R = 16, 512
    g = torch.Generator(device="cpu").manual_seed(3)
    roi = [64, 128, 192, 128, 0, 256, 256, 192]
    src = torch.randn(3 * C, roi[7], roi[6], generator=g).to(cuda).half()
    sp = _disc_spans(R, [(roi[p] + roi[6] // 2, roi[3 + p] + roi[7] // 2) for p in range(3)], 80, np.random.default_rng(7))
    spd = torch.from_numpy(sp).to(cuda)
    res = []
    for spans in (None, spd):
        tm = torch.full((3, R, R, C), 7.0, dtype=torch.float16, device=cuda)
        L.check(lib.tnl_planes_half_to_texel_major_spans(L.ptr(src), L.u32(C), L.u32(R), L.ptr(tm),
                                                         L.roi_array(roi + [C, 0]), L.ptr(spans), L.stream()), "layout")
        res.append(tm)
    mask = _piece_mask(sp, R, 1, cuda)
    for p in range(3):
        inside = torch.zeros(R, R, dtype=torch.bool, device=cuda)
        inside[roi[3 + p]:roi[3 + p] + roi[7], roi[p]:roi[p] + roi[6]] = True
        a, b = res[0][p], res[1][p]
        mk = mask[p] & inside
        assert torch.equal(a[mk], b[mk]) and int(mk.sum()) > 5000
        assert bool(((a == b).all(-1) | (b == 7.0).all(-1)).all())
        assert int((b[inside] == 7.0).all(-1).sum()) > 0.2 * roi[6] * roi[7]


def test_row_extents_kernel_against_the_definition(cuda):
    """tnl_occupancy_row_extents against a host loop over the occupied cells of a two-cascade shell."""
    import trinerflet_amd._lib as L
    from oracle import cref
    Hg, casc, bound, R = 32, 2, 1.5, 256
    bf = synthetic.sphere_bitfield(Hg, casc, bound, 0.7, 0.3)
    bits = torch.from_numpy(bf).to(cuda).view(casc, -1)
    ext = torch.tensor([BIG, -1], dtype=torch.int32, device=cuda).repeat(3 * (R // 8))
    L.check(L.lib().tnl_occupancy_row_extents(L.ptr(bits), L.u32(bits.shape[1]), L.u32(casc), L.u32(Hg), L.f32(bound),
                                              L.u32(R), L.ptr(ext), L.stream()), "row_extents")
    got = ext.cpu().numpy().reshape(3, R // 8, 2)
    want = np.empty_like(got)
    want[..., 0], want[..., 1] = BIG, -1
    unpacked = np.unpackbits(bf, bitorder="little").reshape(casc, Hg ** 3)
    f32 = np.float32
    for c in range(casc):
        sc = f32(min(2.0 ** c, bound))
        cells = np.nonzero(unpacked[c])[0]
        for xyz in cref.morton3D_invert(cells):
            t0, t1 = [], []
            for a in range(3):
                w0 = (f32(xyz[a]) / f32(Hg) * f32(2) - f32(1)) * sc
                w1 = (f32(xyz[a] + 1) / f32(Hg) * f32(2) - f32(1)) * sc
                f0 = (np.clip(w0 / f32(bound), f32(-1), f32(1)) + f32(1)) * f32(0.5) * f32(R - 1)
                f1 = (np.clip(w1 / f32(bound), f32(-1), f32(1)) + f32(1)) * f32(0.5) * f32(R - 1)
                t0.append(max(int(np.floor(f0)) - 1, 0))
                t1.append(min(int(np.floor(f1)) + 3, R))
            for p, (xa, ya) in enumerate(((0, 2), (0, 1), (1, 2))):
                for gi in range(t0[ya] >> 3, ((t1[ya] - 1) >> 3) + 1):
                    want[p, gi, 0] = min(want[p, gi, 0], t0[xa])
                    want[p, gi, 1] = max(want[p, gi, 1], t1[xa])
    assert np.array_equal(got, want)
    assert (got[..., 1] > got[..., 0]).sum() > 20 and (got[..., 1] < 0).sum() > 0


def _setup(cuda, N=2048, bound=1.0, radius=0.3, C=16, R=1024, scale=16):
    from trinerflet_amd.nerf.network import NeRFNetwork
    o, d = synthetic.training_rays(N, n_cams=4, seed=7)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    gt = t(synthetic.target_colors(d))
    noise = t(np.random.default_rng(0).random(N).astype(np.float32))
    m = NeRFNetwork(encoding="triplane_wavelet", bound=bound, cuda_ray=True, density_thresh=10, hidden_dim=64,
                    hidden_dim_color=64, triplane_channels=C, triplane_resolution=R, triplane_wavelet_levels=scale,
                    wavelet_type="bior6.8").to(cuda)
    synthetic.init_field_parameters(m, seed=3)
    bf = t(synthetic.sphere_bitfield(128, 1, bound, radius, 0.0))
    m.density_bitfield.copy_(bf)
    return t(o), t(d), gt, noise, m, bf


def test_training_with_the_pieces_is_the_same_training(cuda):
    """Ordered plane-gradient reduction (deterministic=True): ten steps (refreshes at 0, 4, 8) with the pieces -- forward
    levels, layout change, adjoint and optimiser pass all restricted -- and without.  Every step's rendered colours and,
    after the replay, every parameter and both Adam moments: the same bits.  R = 1024: the finest level runs the
    column-walk kernels, the ones that skip."""
    from trinerflet_amd.train import TrainStep
    o, d, gt, noise, base, bf = _setup(cuda)
    res = []
    for bands in (False, True):
        m = copy.deepcopy(base)
        ts = TrainStep(m, update_extra_interval=4, use_roi=True, defer_adam=True, deterministic=True, live_bands=bands)
        ts.post_refresh = lambda m=m: m.density_bitfield.copy_(bf)
        m.mean_count = 0
        images, used = [], 0
        for it in range(10):
            ts.step(o, d, gt, noises=noise)
            images.append(ts.last["image"].clone())
            if bands and it % 4 >= 1:
                fw, plane = ts._forward_spans()
                assert plane is not None and fw[ts.J - 1] is not None
                used += sum(t is not None for t in ts._adjoint_spans())
        ts.flush_deferred()
        res.append((images, [p.detach().clone() for p in m.parameters()], ts.coef.m.clone(), ts.coef.v.clone(), used))
    assert res[1][4] >= 6                                             # the adjoint ran restricted, too
    for k, (a, b) in enumerate(zip(res[0][0], res[1][0])):
        assert torch.equal(a, b), ("step", k, float((a - b).abs().max()))
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b), float((a - b).abs().max())
    assert torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3])
