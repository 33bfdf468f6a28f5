"""Occupancy-window (ROI) variants of the dense kernels: each must equal its whole-plane counterpart restricted
to the window, and a training run with the window must equal the run without it.

The whole-plane kernels are the ones pinned to the oracle / golden vectors (tests/test_triplane_gpu.py,
tests/test_field_gpu.py); here the ROI entry points are tied to them bit for bit, and the window itself is
checked against positions drawn from the C oracle's march (every footprint must fall inside).
"""
import copy

import numpy as np
import pytest
import torch

from oracle import cref
from trinerflet_amd import synthetic

pytestmark = pytest.mark.gpu


def _roi10(ox, oy, rw, rh, C, s0=0):
    return list(ox) + list(oy) + [rw, rh, C, s0]


@pytest.mark.parametrize("wave", ["bior6.8", "bior2.2", "haar"])
def test_forward_and_adjoint_roi_match_whole_plane(cuda, wave):
    from trinerflet_amd import _lib as L
    from trinerflet_amd.triplaneencoder import triplane_encoder as te
    C, n = 8, 128
    R = 2 * n
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(3, C, n, n, generator=g).to(cuda)
    yh = torch.randn(3, C, 3, n, n, generator=g).to(cuda)
    wid = te.WAVELET_IDS[wave]
    full = te.idwt_level_half(x, yh, wid)                                # (3,C,R,R) fp16
    ox, oy, rw, rh = (64, 0, 128), (128, 64, 0), 128, 64
    roi = _roi10(ox, oy, rw, rh, C)
    comp = te.idwt_level_half_roi(x, yh, wid, roi)                       # (3C, rh, rw)
    for p in range(3):
        ref = full[p, :, oy[p]:oy[p] + rh, ox[p]:ox[p] + rw]
        assert torch.equal(comp.view(3, C, rh, rw)[p], ref), (wave, p)
    # layout change into the persistent texel-major array: window replaced, rest untouched
    tm = torch.full((3, R, R, C), 7.0, dtype=torch.float16, device=cuda)
    te.half_roi_into_texel_major(comp, tm, roi)
    tm_full = te.half_to_texel_major(full)
    for p in range(3):
        win = (slice(oy[p], oy[p] + rh), slice(ox[p], ox[p] + rw))
        assert torch.equal(tm[p][win], tm_full[p][win])
        mask = torch.ones(R, R, dtype=torch.bool, device=cuda)
        mask[win] = False
        assert bool((tm[p][mask] == 7.0).all())
    # adjoint: compact gradient == zero-extended gradient through the whole-plane adjoint, bit for bit
    gc = torch.randn(3 * C, rh, rw, generator=g).to(cuda)
    gfull = torch.zeros(3, C, R, R, device=cuda)
    for p in range(3):
        gfull[p, :, oy[p]:oy[p] + rh, ox[p]:ox[p] + rw] = gc.view(3, C, rh, rw)[p]
    lib = L.lib()
    outs = []
    for use_roi in (False, True):
        dx = torch.full((3 * C, n, n), float("nan"), device=cuda)
        dyh = torch.full((3 * C, 3, n, n), float("nan"), device=cuda)
        src = gc if use_roi else gfull
        L.check(lib.tnl_idwt_level_backward_roi(L.ptr(src), L.u32(3 * C), L.u32(n), L.i32(wid), L.ptr(dx), L.ptr(dyh),
                                                L.roi_array(roi) if use_roi else None, L.stream()), "bwd")
        outs.append((dx, dyh))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # a rank owning slices [s0, s1): same numbers as the corresponding rows of the whole call
    s0, s1 = C + 2, 2 * C + 3
    sub = te.idwt_level_half_roi(x.view(1, 3 * C, n, n)[:, s0:s1], yh.view(1, 3 * C, 3, n, n)[:, s0:s1], wid,
                                 _roi10(ox, oy, rw, rh, C, s0))
    assert torch.equal(sub, comp[s0:s1])


def test_plane_gradient_roi_matches_whole_plane(cuda):
    from trinerflet_amd.nerf import field as F_
    C, R, M, bound = 16, 256, 20000, 1.0
    rng = np.random.default_rng(5)
    # positions inside a box whose footprint is covered by the window below
    xyz = np.stack([rng.uniform(-0.45, 0.2, M), rng.uniform(-0.3, 0.45, M), rng.uniform(0.05, 0.45, M)], 1)
    xyz = torch.from_numpy(xyz.astype(np.float32)).to(cuda)
    dfeat = torch.from_numpy(rng.standard_normal((3, M, C)).astype(np.float16)).to(cuda)
    # texel = (u+1)/2*255: x in [70,153] -> [64,192), y in [89,185] -> [64,192), z in [133,185] -> [128,192)
    ox, oy, rw, rh = (64, 64, 64), (64, 64, 64), 128, 128   # plane0 (x,z), plane1 (x,y), plane2 (y,z)
    full = torch.empty(3, C, R, R, device=cuda)
    F_.plane_grad_binned(dfeat, xyz, bound, C, R, full, channel_major=True)
    comp = torch.full((3 * C, rh, rw), float("nan"), device=cuda)
    F_.plane_grad_binned(dfeat, xyz, bound, C, R, comp, channel_major=True, roi=_roi10(ox, oy, rw, rh, C))
    comp = comp.view(3, C, rh, rw)
    tot = 0.0
    for p in range(3):
        ref = full[p, :, oy[p]:oy[p] + rh, ox[p]:ox[p] + rw]
        # same records per tile, order inside a tile set by the bin-fill atomics: equal up to fp32 summation order
        assert torch.allclose(comp[p], ref, rtol=1e-4, atol=1e-4), p
        tot += float(ref.abs().sum())
    assert abs(float(full.abs().sum()) - tot) <= 1e-6 * tot   # nothing of the gradient lies outside the window


def test_support_chain_adjoint_and_rect_adam_match_dense(cuda):
    """Windowed adjoint over two levels (compact window -> rectangle -> strided window -> rectangle) equals the
    whole-plane adjoint of the zero-extended gradient inside the rectangles and leaves everything outside untouched;
    Adam that takes g = 0 outside a rectangle equals plain Adam on the zero-filled gradient, bit for bit."""
    import ctypes
    from trinerflet_amd import _lib as L
    lib = L.lib()
    C, n1, wid = 8, 64, 4                     # levels: 64 -> 128 -> 256 (fine)
    S, R = 3 * C, 4 * n1
    g = torch.Generator(device="cpu").manual_seed(2)
    ox, oy, rw, rh = (64, 0, 128), (128, 64, 0), 64, 128
    gc = torch.randn(S, rh, rw, generator=g).to(cuda)
    gfull = torch.zeros(3, C, R, R, device=cuda)
    for p in range(3):
        gfull[p, :, oy[p]:oy[p] + rh, ox[p]:ox[p] + rw] = gc.view(3, C, rh, rw)[p]
    # dense reference chain
    ref = []
    src = gfull.view(S, R, R)
    for n in (2 * n1, n1):
        dx = torch.empty(S, n, n, device=cuda)
        dyh = torch.empty(S, 3, n, n, device=cuda)
        L.check(lib.tnl_idwt_level_backward(L.ptr(src), L.u32(S), L.u32(n), L.i32(wid), L.ptr(dx), L.ptr(dyh),
                                            L.stream()), "bwd")
        ref.append((dx, dyh))
        src = dx
    # windowed chain
    win, strided, src = _roi10(ox, oy, rw, rh, C), 0, gc
    out = []
    for lvl, n in enumerate((2 * n1, n1)):
        dx = torch.full((S, n, n), 123.0, device=cuda)
        dyh = torch.full((S, 3, n, n), 123.0, device=cuda)
        rect = (ctypes.c_int32 * 8)()
        L.check(lib.tnl_idwt_level_backward_win(L.ptr(src), L.u32(S), L.u32(n), L.i32(wid), L.ptr(dx), L.ptr(dyh),
                                                L.roi_array(win), L.i32(strided), rect, L.stream()), "bwd_win")
        rect = list(rect)
        out.append((dx, dyh, rect))
        assert rect[6] % 8 == 0 and rect[7] % 8 == 0 and (rect[6] < n or rect[7] < n or lvl == 1)   # 32 (tile kernels) or 8 (column walk)
        for p in range(3):
            ys, xs = slice(rect[3 + p], rect[3 + p] + rect[7]), slice(rect[p], rect[p] + rect[6])
            sl = slice(p * C, (p + 1) * C)
            assert torch.equal(dx[sl, ys, xs], ref[lvl][0][sl, ys, xs])
            assert torch.equal(dyh[sl, :, ys, xs], ref[lvl][1][sl, :, ys, xs])
            mask = torch.ones(n, n, dtype=torch.bool, device=cuda)
            mask[ys, xs] = False
            assert bool((dx[sl][:, mask] == 123.0).all()) and bool((dyh[sl][:, :, mask] == 123.0).all())
            # ... and the dense result really is zero there
            assert float(ref[lvl][0][sl][:, mask].abs().sum()) == 0 and float(ref[lvl][1][sl][:, :, mask].abs().sum()) == 0
        win, strided, src = rect + [C, 0], 1, dx
    # Adam: rectangle-aware on the stale-outside gradient == plain on the dense gradient
    dyh_w, rect = out[0][1], out[0][2]
    n = 2 * n1
    p0 = torch.randn(S * 3 * n * n, generator=g).to(cuda)
    m0 = torch.randn(S * 3 * n * n, generator=g).to(cuda) * 1e-3
    v0 = torch.rand(S * 3 * n * n, generator=g).to(cuda) * 1e-6
    steps = torch.full((1,), 4.0, device=cuda)
    found = torch.zeros(1, device=cuda)
    res = []
    for use_rect in (False, True):
        p_, m_, v_ = p0.clone(), m0.clone(), v0.clone()
        gbuf = (dyh_w if use_rect else ref[0][1]).reshape(-1).clone()
        abs_sum = torch.zeros(1, device=cuda)
        if use_rect:
            L.check(lib.tnl_adam_l1_step_rect(L.ptr(p_), L.ptr(gbuf), L.ptr(m_), L.ptr(v_), L.u32(S), L.u32(3), L.u32(n),
                                              L.u32(C), L.u32(0), (ctypes.c_int32 * 8)(*rect), L.f32(1e-2), L.ptr(steps),
                                              L.f32(0.9), L.f32(0.99), L.f32(1e-15), L.f32(0.5), None, L.f32(1e-4),
                                              L.ptr(found), L.ptr(abs_sum), L.stream()), "adam_rect")
        else:
            L.check(lib.tnl_adam_l1_step_dev(L.ptr(p_), L.ptr(gbuf), L.ptr(m_), L.ptr(v_), L.u64(p_.numel()), L.f32(1e-2),
                                             L.ptr(steps), L.f32(0.9), L.f32(0.99), L.f32(1e-15), L.f32(0.5), None,
                                             L.f32(1e-4), L.ptr(found), L.ptr(abs_sum), L.i32(0), L.stream()), "adam")
        res.append((p_, m_, v_, abs_sum))
    for k, (a, b) in enumerate(zip(*res)):
        if k < 3:
            assert torch.equal(a, b), ("pmv"[k], float((a - b).abs().max()), int((a != b).sum()))
        else:
            assert torch.allclose(a, b, rtol=1e-5)      # sum of |p|: float atomics, order-dependent
    # a rank that owns slices [s0, s1): same rows
    s0, s1 = C - 2, 2 * C + 1
    per = 3 * n * n
    p_, m_, v_ = (t[s0 * per:s1 * per].clone() for t in (p0, m0, v0))
    gbuf = dyh_w.reshape(-1)[s0 * per:s1 * per].clone()
    L.check(lib.tnl_adam_l1_step_rect(L.ptr(p_), L.ptr(gbuf), L.ptr(m_), L.ptr(v_), L.u32(s1 - s0), L.u32(3), L.u32(n),
                                      L.u32(C), L.u32(s0), (ctypes.c_int32 * 8)(*rect), L.f32(1e-2), L.ptr(steps),
                                      L.f32(0.9), L.f32(0.99), L.f32(1e-15), L.f32(0.5), None, L.f32(1e-4), L.ptr(found),
                                      None, L.stream()), "adam_rect")
    assert torch.equal(p_, res[0][0][s0 * per:s1 * per]) and torch.equal(v_, res[0][2][s0 * per:s1 * per])


def _model(dev, C=16, R=256, scale=4, H=64, bound=1.0, **kw):
    from trinerflet_amd.nerf.network import NeRFNetwork
    m = NeRFNetwork(encoding="triplane_wavelet", bound=bound, cuda_ray=True, density_thresh=10, hidden_dim=H,
                    hidden_dim_color=H, triplane_channels=C, triplane_resolution=R, triplane_wavelet_levels=scale,
                    wavelet_type="bior6.8", **kw).to(dev)
    synthetic.init_field_parameters(m, seed=3)
    return m


def test_window_covers_every_marched_footprint(cuda):
    """The window computed from the bitfield contains the bilinear footprint of every sample the ORACLE's march
    produces for that bitfield (off-centre occupancy, two cascades)."""
    from trinerflet_amd.train import TrainStep
    bound, R, Hg = 2.0, 256, 128
    m = _model(cuda, R=R, bound=bound)
    # off-centre blob: occupied cells of cascade 0 with centre within 0.35 of (0.3, -0.2, 0.1); cascade 1 likewise
    idx = np.arange(Hg)
    bits = np.zeros((2, Hg ** 3), np.uint8)
    coords = np.stack(np.meshgrid(idx, idx, idx, indexing="ij"), -1).reshape(-1, 3)
    mort = cref.morton3D(coords.astype(np.uint32))
    for cas in range(2):
        b = min(2.0 ** cas, bound)
        c = ((coords + 0.5) / Hg * 2 - 1) * b
        occ = np.linalg.norm(c - np.array([0.3, -0.2, 0.1]), axis=1) < 0.35
        bits[cas, mort] = occ
    bf = np.packbits(bits.reshape(-1, 8), axis=1, bitorder="little").reshape(-1)
    m.density_bitfield.copy_(torch.from_numpy(bf).to(cuda))
    ts = TrainStep(m, update_extra_interval=0)
    roi = ts._compute_roi()
    assert roi is not None and roi[6] < R and roi[7] < R
    N = 4096
    o, d = synthetic.training_rays(N, n_cams=6, seed=2)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = cref.near_far_from_aabb(o, d, aabb, 0.2)
    noise = np.random.default_rng(1).random(N).astype(np.float32)
    xyzs, _, _, _, counter = cref.march_rays_train(o, d, bound, bf, 2, Hg, nears, fars, noise, N * 64)
    n_samples = int(counter[0])
    assert 1000 < n_samples <= N * 64
    u = np.clip(xyzs[:n_samples] / bound, -1, 1)
    f = (u + 1) / 2 * (R - 1)
    t0 = np.floor(f).astype(int)
    t1 = np.minimum(t0 + 1, R - 1)
    xa, ya = (0, 0, 1), (2, 1, 2)
    for p in range(3):
        assert t0[:, xa[p]].min() >= roi[p] and t1[:, xa[p]].max() < roi[p] + roi[6]
        assert t0[:, ya[p]].min() >= roi[3 + p] and t1[:, ya[p]].max() < roi[3 + p] + roi[7]


@pytest.mark.parametrize("plane_dtype", [torch.float16, torch.float32])
def test_windowed_rebuild_equals_full_rebuild_inside_the_window(cuda, plane_dtype):
    """Three wavelet levels: the window chain (every level only computes what the next one needs) reproduces the full
    rebuild inside the occupancy window bit for bit and leaves the rest of the persistent array alone.  fp32 planes (the
    reference's training precision, utils.py:1138-1140; windowed since round 6): the finest level writes its window of a
    full-size fp32 array, the layout pass converts that window."""
    from trinerflet_amd.train import TrainStep
    m = _model(cuda, R=512, scale=8, bound=1.0, plane_dtype=plane_dtype)
    with torch.no_grad():
        for p in m.encoder.planes_features_wavelet_coefs:
            p.normal_(0, 0.05)
    m.density_bitfield.copy_(torch.from_numpy(synthetic.sphere_bitfield(128, 1, 1.0, 0.12, 0.0)).to(cuda))
    ts = TrainStep(m, update_extra_interval=0, live_bands=False)   # the whole window (pieces: tests/test_spans_gpu.py)
    full = ts.rebuild_planes().clone()                      # whole planes, sets the persistent array
    ts._roi, ts._roi_valid = ts._compute_roi(), True
    roi = ts._roi
    assert roi is not None and roi[6] <= 128
    wins = ts._forward_windows()
    assert wins[2] == list(roi) and wins[1] is not None, (roi, wins)     # the level below the finest is windowed too
    ts._tm_full.fill_(5.0)
    tm = ts.rebuild_planes(roi=True)
    assert tm.dtype == plane_dtype and tm is ts._tm_full
    for p in range(3):
        win = (slice(roi[3 + p], roi[3 + p] + roi[7]), slice(roi[p], roi[p] + roi[6]))
        assert torch.equal(tm[p][win], full[p][win])
        mask = torch.ones(512, 512, dtype=torch.bool, device=cuda)
        mask[win] = False
        assert bool((tm[p][mask] == 5.0).all())


@pytest.mark.parametrize("plane_dtype", [torch.float16, torch.float32])
def test_training_with_window_equals_whole_plane_training(cuda, plane_dtype):
    """Six steps (one grid refresh inside) with and without the occupancy window: same parameters."""
    from trinerflet_amd.train import TrainStep
    N, bound = 2048, 1.0
    o, d = synthetic.training_rays(N, n_cams=4, seed=7)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    gt = t(synthetic.target_colors(d))
    noise = t(np.random.default_rng(0).random(N).astype(np.float32))
    base = _model(cuda, bound=bound, plane_dtype=plane_dtype)
    bf = t(synthetic.sphere_bitfield(128, 1, bound, 0.4, 0.0))
    base.density_bitfield.copy_(bf)
    res = []
    for use_roi in (False, True):
        m = copy.deepcopy(base)
        ts = TrainStep(m, update_extra_interval=4, use_roi=use_roi)
        assert m.encoder.plane_dtype == plane_dtype
        ts.post_refresh = lambda m=m: m.density_bitfield.copy_(bf)   # keep the analytic occupancy
        m.mean_count = 0
        losses = []
        for it in range(6):
            losses.append(float(ts.step(t(o), t(d), gt, noises=noise)))
            if use_roi and it % 4 != 0:
                assert ts._roi is not None and ts._roi[6] < 256
        res.append((losses, [p.detach().clone() for p in m.parameters()], ts))
    assert res[1][2].use_roi and res[1][2]._roi is not None
    assert res[1][2]._rect_ok and all(r is not None for r in res[1][2]._rects)   # the support chain was active
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=2e-4)
    for a, b in zip(res[0][1], res[1][1]):
        # The tile reduction sums in the order the bin-fill atomics produced, so gradients differ in the last bits
        # from run to run; Adam (eps = 1e-15) turns a gradient at noise level into a +-lr step.  Such coefficients
        # are isolated and always the same few candidates (where the gradient nearly cancels): 0-9 of 0.6 M were
        # seen to flip between two runs; allow a 1e-4 fraction of them, bounded by lr * steps.
        bad = ((a - b).abs() > 2e-3 + 1e-3 * b.abs())
        assert int(bad.sum()) <= max(8, int(1e-4 * a.numel())) and float((a - b).abs().max()) < 6e-2, \
            (int(bad.sum()), a.numel(), float((a - b).abs().max()), bad.nonzero()[:5].tolist())


@pytest.mark.gpu
def test_occupancy_bounds_kernel(cuda):
    """tnl_occupancy_bounds against the plain definition (decode every set bit of the Morton-ordered bitfield)."""
    import trinerflet_amd._lib as L
    from trinerflet_amd import raymarching
    Hg, casc = 32, 3
    g = torch.Generator().manual_seed(5)
    grid = torch.zeros(casc, Hg ** 3)
    idx = torch.randint(0, Hg ** 3, (40,), generator=g)
    grid[0, idx] = 1.0
    grid[2, 12345] = 1.0                                   # cascade 1 stays empty
    bits = raymarching.packbits(grid.to(cuda), 0.5)
    bounds = torch.tensor([[Hg + 1] * 3 + [-1] * 3] * casc, dtype=torch.int32, device=cuda)
    L.check(L.lib().tnl_occupancy_bounds(L.ptr(bits), L.u32(Hg ** 3 // 8), L.u32(casc), L.ptr(bounds), L.stream()), "ob")
    coords = raymarching.morton3D_invert(torch.arange(Hg ** 3, dtype=torch.int32, device=cuda)).cpu()
    for c in range(casc):
        cells = coords[grid[c] > 0]
        want = ([Hg + 1] * 3 + [-1] * 3) if len(cells) == 0 else cells.amin(0).tolist() + cells.amax(0).tolist()
        assert bounds[c].tolist() == want


def test_side_work_beside_the_step_does_not_change_the_training(cuda):
    """The next batch's march + tile sort run on a side stream underneath the step's gradient / optimiser kernels
    (step(next_rays=...)), or in order on the launch stream (overlap_march = False, no next_rays): the training run must
    not change.  Also the far clip of the in-order march of refresh steps (clip_far_in_order: the same samples)."""
    from trinerflet_amd.train import TrainStep
    N, bound = 2048, 1.0
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    batches = []
    for seed in (7, 8):
        o, d = synthetic.training_rays(N, n_cams=4, seed=seed)
        batches.append((t(o), t(d), t(synthetic.target_colors(d)),
                        t(np.random.default_rng(seed).random(N).astype(np.float32))))
    base = _model(cuda, bound=bound)
    bf = t(synthetic.sphere_bitfield(128, 1, bound, 0.4, 0.0))
    base.density_bitfield.copy_(bf)
    res = {}
    for pf in ("side", "in_order", "no_clip", "side2"):   # side2: the same configuration again, the yardstick of run-to-run noise
        m = copy.deepcopy(base)
        ts = TrainStep(m, update_extra_interval=4)
        ts.overlap_march = pf != "in_order"
        ts.clip_far_in_order = pf != "no_clip"
        ts.post_refresh = lambda m=m: m.density_bitfield.copy_(bf)
        m.mean_count = 0
        losses, counts = [], []
        for it in range(7):
            o, d, gt, nz = batches[it % 2]
            no, nd, _, nnz = batches[(it + 1) % 2]
            losses.append(float(ts.step(o, d, gt, noises=nz, next_rays=None if pf == "in_order" else (no, nd, nnz))))
            counts.append(int(ts.last["counter"][0]))
        torch.cuda.synchronize()
        res[pf] = (losses, [p.detach().clone() for p in m.parameters()], counts)
    for pf in ("in_order", "no_clip"):
        assert res[pf][2] == res["side"][2], pf                     # the same samples
        np.testing.assert_allclose(res["side"][0], res[pf][0], rtol=2e-4, err_msg=pf)
        for a, b, c in zip(res["side"][1], res[pf][1], res["side2"][1]):
            # see test_training_with_window_equals_whole_plane_training: Adam turns a gradient at the tile
            # reduction's noise level (its summation order follows the sort's atomics) into +-lr steps; two runs of
            # the SAME configuration differ that way too, and that is the measure
            far = lambda x, y: int(((x - y).abs() > 2e-3 + 1e-3 * y.abs()).sum())
            assert far(b, a) <= 10 * far(c, a) + max(8, int(2e-4 * a.numel())), (pf, far(b, a), far(c, a))   # the count scatters
            assert float((a - b).abs().max()) < 2 * 7 * 1e-2, pf


def test_banded_plane_gradient_equals_the_single_reduction_bit_for_bit(cuda):
    """TrainStep(overlap_exchange=K): the plane-gradient window reduced in K bands of rows (each band's tiles in a launch
    of its own, the results concatenated) is the same array as the single reduction -- with the ordered tile lists
    (deterministic=True) the whole training run agrees to the bit (one process: no collective, only the banding)."""
    from trinerflet_amd.train import TrainStep
    N, bound = 4096, 1.0
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    o, d = synthetic.training_rays(N, n_cams=4, seed=7)
    gt, nz = synthetic.target_colors(d), np.random.default_rng(1).random(N).astype(np.float32)
    base = _model(cuda, R=512, scale=8, bound=bound)
    bf = t(synthetic.sphere_bitfield(128, 1, bound, 0.5, 0.0))
    base.density_bitfield.copy_(bf)
    res = []
    for K in (0, 3):
        m = copy.deepcopy(base)
        ts = TrainStep(m, update_extra_interval=4, deterministic=True, overlap_exchange=K)
        ts.post_refresh = lambda m=m: m.density_bitfield.copy_(bf)
        m.mean_count = 0
        losses = [float(ts.step(t(o), t(d), t(gt), noises=t(nz))) for _ in range(6)]
        assert ts._roi is not None and (K == 0 or len(ts._exchange_bands(ts._roi)) == K)
        ts.flush_deferred()
        res.append((losses, [p.detach().clone() for p in m.parameters()]))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=2e-6)     # (the reported MSE is a float-atomic sum)
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
