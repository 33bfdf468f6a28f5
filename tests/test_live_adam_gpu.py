"""tnl_idwt_level_backward_live_adam at the C ABI: a column-walk level of the windowed adjoint with the optimiser's live
pass in its epilogue must leave exactly the bits of the two calls it replaces -- tnl_idwt_level_backward_spans (the band
gradients into a buffer) followed by tnl_adam_l1_step_live_bands over the same live rectangle and step record -- in p, m, v
and in the low-pass gradient it hands to the next level; nothing outside the live rectangle moves; a step GradScaler skips
moves nothing at all.  (Autograd's backward of triplane_encoder.py:392-394 + reconstruction/nerf/utils.py:1166-1173.)"""
import ctypes as C_

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def walk_levels(cuda):
    from trinerflet_amd import _lib as L
    L.lib().tnl_idwt_set_walk_min_n(L.u32(32))
    yield
    L.lib().tnl_idwt_set_walk_min_n(L.u32(0))


def _i32(vals):
    return (C_.c_int32 * len(vals))(*[int(v) for v in vals])


@pytest.mark.parametrize("grow", [0, 8], ids=["live-is-the-support", "live-wider-than-the-support"])
@pytest.mark.parametrize("skip", [False, True], ids=["step", "skipped-step"])
def test_fused_level_equals_adjoint_then_live_pass(cuda, walk_levels, grow, skip):
    from trinerflet_amd import _lib as L
    lib = L.lib()
    C, n, wave = 2, 64, 4                      # 3 * C slices of a 64 x 64 level (bior6.8), fine grid 128 x 128
    S, m2 = 3 * C, 2 * n
    g = torch.Generator().manual_seed(7)
    win = [64, 0, 64, 0, 64, 64, 64, 64]      # per-plane origins of a 64 x 64 window of the fine gradient (compact input)
    dout = torch.randn(S, win[7], win[6], generator=g).to(cuda)
    p0 = (torch.randn(S, 3, n, n, generator=g) * 0.1).to(cuda)
    m0 = (torch.randn(S, 3, n, n, generator=g) * 0.01).to(cuda)
    v0 = (torch.rand(S, 3, n, n, generator=g) * 1e-3).to(cuda)
    inv_scale = torch.tensor([1.0 / 1024.0], device=cuda)
    found_inf = torch.tensor([1.0 if skip else 0.0], device=cuda)
    opt_steps = torch.tensor([5.0], device=cuda)
    ring = torch.zeros(16 * 4, device=cuda)
    lr, b1, b2, eps, l1 = 1e-2, 0.9, 0.99, 1e-15, 3e-4
    L.check(lib.tnl_adam_record_step(L.ptr(ring), L.i32(3), L.f32(lr), L.ptr(opt_steps), L.f32(b1), L.f32(b2),
                                     L.ptr(found_inf), L.stream()), "record")
    rec = ring[12:]
    roi = L.roi_array(win + [C, 0])
    # the separate passes
    dx_a = torch.zeros(S, n, n, device=cuda)
    dyh = torch.zeros(S, 3, n, n, device=cuda)
    rect = (C_.c_int32 * 8)()
    L.check(lib.tnl_idwt_level_backward_spans(L.ptr(dout), L.u32(S), L.u32(n), L.i32(wave), L.ptr(dx_a), L.ptr(dyh), roi,
                                              L.i32(0), rect, L.ptr(None), L.stream()), "adjoint")
    rect = list(rect)
    assert rect[6] % 8 == 0 and rect[7] % 8 == 0 and 0 < rect[6] < n
    lw, lh = min(rect[6] + 2 * grow, n), min(rect[7] + 2 * grow, n)
    live = [min(max(rect[k] - grow, 0), n - lw) for k in range(3)] + [min(max(rect[3 + k] - grow, 0), n - lh) for k in range(3)] \
        + [lw, lh]
    pa, ma, va = p0.clone(), m0.clone(), v0.clone()
    sum_a = torch.zeros(1, device=cuda)
    L.check(lib.tnl_adam_l1_step_live_bands(
        L.ptr(pa), L.ptr(dyh), L.ptr(ma), L.ptr(va), L.u32(S), L.u32(C), L.u32(0), L.u32(1), (C_.c_uint64 * 1)(0),
        (C_.c_uint32 * 1)(n), (C_.c_uint32 * 1)(3), _i32(live), _i32(rect), (C_.c_void_p * 1)(None), (C_.c_uint32 * 1)(0),
        (C_.c_float * 1)(l1), L.f32(lr), L.ptr(opt_steps), L.ptr(rec), L.f32(b1), L.f32(b2), L.f32(eps), L.f32(1.0),
        L.ptr(inv_scale), L.ptr(found_inf), L.ptr(sum_a), L.stream()), "live pass")
    # the fused level
    pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
    dx_b = torch.zeros(S, n, n, device=cuda)
    sum_b = torch.zeros(1, device=cuda)
    L.check(lib.tnl_idwt_level_backward_live_adam(
        L.ptr(dout), L.u32(S), L.u32(n), L.i32(wave), L.ptr(dx_b), roi, L.i32(0), _i32(live), L.ptr(None), L.ptr(pb),
        L.ptr(mb), L.ptr(vb), L.ptr(None), L.u32(0), L.ptr(rec), L.f32(b1), L.f32(b2), L.f32(eps), L.f32(1.0),
        L.ptr(inv_scale), L.f32(l1), L.ptr(found_inf), L.ptr(sum_b), L.stream()), "fused level")
    torch.cuda.synchronize()
    for a, b in ((pa, pb), (ma, mb), (va, vb)):
        assert torch.equal(a, b)
    inside = torch.zeros(S, 3, n, n, dtype=torch.bool, device=cuda)
    for s in range(S):
        pl = s // C
        inside[s, :, live[3 + pl]:live[3 + pl] + lh, live[pl]:live[pl] + lw] = True
    assert torch.equal(pb[~inside], p0[~inside]) and torch.equal(mb[~inside], m0[~inside])     # nothing outside the live rectangle
    if skip:
        assert torch.equal(pb, p0) and torch.equal(mb, m0) and torch.equal(vb, v0)
    else:
        assert not torch.equal(pb[inside], p0[inside])
    for s in range(S):                                     # the low-pass gradient over the support (what the next level reads)
        pl = s // C
        ys, xs = slice(rect[3 + pl], rect[3 + pl] + rect[7]), slice(rect[pl], rect[pl] + rect[6])
        assert torch.equal(dx_a[s, ys, xs], dx_b[s, ys, xs])
    np.testing.assert_allclose(float(sum_b), float(sum_a), rtol=1e-5)      # sum |p| of the live pieces (float atomics)


def test_fused_level_refuses_what_it_cannot_do(cuda, walk_levels):
    from trinerflet_amd import _lib as L
    lib = L.lib()
    S, n = 6, 64
    z = lambda *sh: torch.zeros(*sh, device=cuda)
    dout, dx, p, m, v, rec = z(S, 64, 64), z(S, n, n), z(S, 3, n, n), z(S, 3, n, n), z(S, 3, n, n), z(4)
    roi = L.roi_array([0, 0, 0, 0, 0, 0, 64, 64, 2, 0])

    def call(live, n_=n, p_=p, rec_=rec):
        return lib.tnl_idwt_level_backward_live_adam(
            L.ptr(dout), L.u32(S), L.u32(n_), L.i32(4), L.ptr(dx), roi, L.i32(0), _i32(live), L.ptr(None), L.ptr(p_), L.ptr(m),
            L.ptr(v), L.ptr(None), L.u32(0), L.ptr(rec_), L.f32(0.9), L.f32(0.99), L.f32(1e-15), L.f32(1.0), L.ptr(None),
            L.f32(0.0), L.ptr(None), L.ptr(None), L.stream())
    ok = [0, 0, 0, 0, 0, 0, 40, 40]
    assert call(ok) == 0
    assert call([2, 0, 0, 0, 0, 0, 40, 40]) != 0          # column origin not a multiple of 4
    assert call([0, 0, 0, 4, 0, 0, 40, 40]) != 0          # row origin not a multiple of 8
    assert call([0, 0, 0, 0, 0, 0, 40, 44]) != 0          # height not a multiple of 8
    assert call([32, 0, 0, 0, 0, 0, 40, 40]) != 0         # leaves the level
    assert call(ok, p_=p.view(-1)[1:]) != 0               # p not 16-byte aligned
    assert call(ok, rec_=None) != 0                       # no step record
    lib.tnl_idwt_set_walk_min_n(L.u32(1 << 20))           # not a column-walk level any more
    try:
        assert call(ok) != 0
    finally:
        lib.tnl_idwt_set_walk_min_n(L.u32(32))
    torch.cuda.synchronize()
