"""The column-walk IDWT / adjoint kernels (csrc/wavelet.hip k_idwt_fwd_walk / k_idwt_bwd_walk; production use: levels
with n >= 512) forced onto small planes with tnl_idwt_set_walk_min_n, so that the C oracle can check them in seconds:
all five wavelets, plane sizes that leave ragged last tiles (n not a multiple of the 120-column tile) and several row
segments, forward bit-identical to the LDS-tiled kernels, adjoint vs the oracle and the <Ax,y> = <x,A^T y> identity,
and the occupancy-window / support-rectangle entry points (the tests of tests/test_roi_gpu.py re-run on these kernels).
The full-size use (n = 512, 1024) is covered by tests/test_full_geometry_gpu.py."""
import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[0, 2, 4], ids=["one-column", "pair-2-rows", "pair-4-rows"])
def fwd_form(request, cuda):
    """Both forms of the forward walk kernel (tnl_idwt_set_tuning key 4: one coarse column per thread / two, with 2 or 4
    coarse rows per phase) run every test of this file: each must reproduce the tile kernels' bits."""
    from trinerflet_amd import _lib as L
    assert L.lib().tnl_idwt_set_tuning(4, request.param) == 0
    yield request.param
    assert L.lib().tnl_idwt_set_tuning(4, -1) == 0      # back to the build's default


@pytest.fixture
def walk(cuda):
    from trinerflet_amd import _lib as L
    L.lib().tnl_idwt_set_walk_min_n(L.u32(8))
    yield
    L.lib().tnl_idwt_set_walk_min_n(L.u32(0))


def _vol(dev, C, R, scale, wave):
    from trinerflet_amd.triplaneencoder.triplane_encoder import TriPlaneVolume
    return TriPlaneVolume(number_of_features=C, plane_resolution=R, inner_multi_res_scale=scale, wavelet_type=wave,
                          plane_dtype=torch.float32).to(dev)


@pytest.mark.parametrize("wave,C,R,scale", [("bior6.8", 3, 512, 4), ("bior6.8", 1, 1008, 2), ("haar", 2, 256, 8),
                                            ("bior4.4", 1, 320, 2), ("bior2.6", 2, 144, 2), ("bior2.2", 2, 64, 4)])
def test_walk_kernels_vs_oracle_and_tile_kernels(cuda, wave, C, R, scale):
    from trinerflet_amd import _lib as L
    torch.manual_seed(0)
    vol = _vol(cuda, C, R, scale, wave)
    with torch.no_grad():
        vol.planes_features.normal_(0, 0.5)
        for p in vol.planes_features_wavelet_coefs:
            p.normal_(0, 0.3)
    res = {}
    cot = None
    for mode, min_n in (("tile", 1 << 20), ("walk", 8)):
        L.lib().tnl_idwt_set_walk_min_n(L.u32(min_n))
        try:
            vol.reset_cahce()
            vol.zero_grad(set_to_none=True)
            planes = vol.get_planes()
            if cot is None:
                cot = torch.randn_like(planes)
            planes.backward(cot)
            res[mode] = (planes.detach().clone(), vol.planes_features.grad.clone(),
                         [p.grad.clone() for p in vol.planes_features_wavelet_coefs])
        finally:
            L.lib().tnl_idwt_set_walk_min_n(L.u32(0))
    # forward: same FMA order per output as the tile kernels -> the same bits
    assert torch.equal(res["walk"][0], res["tile"][0])
    ll = vol.planes_features.detach().cpu().numpy()
    coefs = [p.detach().cpu().numpy() for p in vol.planes_features_wavelet_coefs]
    ref = cref.build_planes(ll, coefs, wave)
    np.testing.assert_allclose(res["walk"][0].cpu().numpy(), ref, rtol=0, atol=3e-6 * np.abs(ref).max())
    # adjoint: vertical pass first (the tile kernels run the horizontal one first): equal to rounding, both vs the oracle
    dll, dcoefs = cref.build_planes_adj(cot.cpu().numpy(), len(coefs), wave)
    np.testing.assert_allclose(res["walk"][1].cpu().numpy(), dll, rtol=0, atol=5e-6 * np.abs(dll).max())
    for g, d, gt in zip(res["walk"][2], dcoefs, res["tile"][2]):
        np.testing.assert_allclose(g.cpu().numpy(), d, rtol=0, atol=5e-6 * np.abs(d).max())
        assert float((g - gt).abs().max()) <= 5e-6 * float(gt.abs().max())
    planes = res["walk"][0]
    lhs = (planes.double() * cot.double()).sum().item()
    rhs = (vol.planes_features.detach().double() * res["walk"][1].double()).sum().item()
    rhs += sum((p.detach().double() * g.double()).sum().item()
               for p, g in zip(vol.planes_features_wavelet_coefs, res["walk"][2]))
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


def test_walk_half_output_equals_tile_kernel(cuda):
    """fp16 output (the finest level's form in TrainStep) and several row segments."""
    from trinerflet_amd import _lib as L
    from trinerflet_amd.triplaneencoder import triplane_encoder as te
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 4, 264, 264, generator=g).to(cuda)
    yh = torch.randn(3, 4, 3, 264, 264, generator=g).to(cuda)
    outs = []
    for min_n in (1 << 20, 8):
        L.lib().tnl_idwt_set_walk_min_n(L.u32(min_n))
        try:
            outs.append(te.idwt_level_half(x, yh, 4))
        finally:
            L.lib().tnl_idwt_set_walk_min_n(L.u32(0))
    assert outs[0].dtype == torch.float16 and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("wave", ["bior6.8", "bior2.2", "haar"])
def test_window_entry_points_on_walk_kernels(cuda, walk, wave):
    from tests import test_roi_gpu as troi
    troi.test_forward_and_adjoint_roi_match_whole_plane(cuda, wave)


def test_support_chain_on_walk_kernels(cuda, walk):
    from tests import test_roi_gpu as troi
    troi.test_support_chain_adjoint_and_rect_adam_match_dense(cuda)
    for plane_dtype in (torch.float16, torch.float32):
        troi.test_windowed_rebuild_equals_full_rebuild_inside_the_window(cuda, plane_dtype)
        troi.test_training_with_window_equals_whole_plane_training(cuda, plane_dtype)


@pytest.mark.parametrize("wave", ["bior6.8", "bior4.4", "haar"])
def test_walk_adjoint_rectangles_random_windows(cuda, walk, wave):
    """The column-walk adjoint writes 8-wide granules of its gradient-support rectangle (the tile kernels 32-wide
    tiles): for random windows, two levels deep (compact window -> rectangle -> strided window -> rectangle), the
    rectangle holds the dense adjoint of the zero-extended gradient bit for bit, the dense result is exactly zero
    outside it, and nothing outside it is written."""
    import ctypes
    from trinerflet_amd import _lib as L
    from trinerflet_amd.triplaneencoder import triplane_encoder as te
    lib = L.lib()
    C, n1 = 4, 64
    S, R = 3 * C, 4 * n1
    wid = te.WAVELET_IDS[wave]
    rng = np.random.default_rng(7)
    g = torch.Generator(device="cpu").manual_seed(3)
    for trial in range(6):
        rw, rh = int(rng.choice([64, 128])), int(rng.choice([64, 128]))
        ox = [int(rng.integers(0, (R - rw) // 64 + 1)) * 64 for _ in range(3)]
        oy = [int(rng.integers(0, (R - rh) // 64 + 1)) * 64 for _ in range(3)]
        gc = torch.randn(S, rh, rw, generator=g).to(cuda)
        gfull = torch.zeros(3, C, R, R, device=cuda)
        for p in range(3):
            gfull[p, :, oy[p]:oy[p] + rh, ox[p]:ox[p] + rw] = gc.view(3, C, rh, rw)[p]
        ref, src = [], gfull.view(S, R, R)
        for n in (2 * n1, n1):
            dx = torch.empty(S, n, n, device=cuda)
            dyh = torch.empty(S, 3, n, n, device=cuda)
            L.check(lib.tnl_idwt_level_backward(L.ptr(src), L.u32(S), L.u32(n), L.i32(wid), L.ptr(dx), L.ptr(dyh),
                                                L.stream()), "bwd")
            ref.append((dx, dyh))
            src = dx
        win, strided, src = ox + oy + [rw, rh, C, 0], 0, gc
        for lvl, n in enumerate((2 * n1, n1)):
            dx = torch.full((S, n, n), 123.0, device=cuda)
            dyh = torch.full((S, 3, n, n), 123.0, device=cuda)
            rect = (ctypes.c_int32 * 8)()
            L.check(lib.tnl_idwt_level_backward_win(L.ptr(src), L.u32(S), L.u32(n), L.i32(wid), L.ptr(dx), L.ptr(dyh),
                                                    L.roi_array(win), L.i32(strided), rect, L.stream()), "bwd_win")
            rect = list(rect)
            assert rect[6] % 8 == 0 and rect[7] % 8 == 0 and all(v % 4 == 0 for v in rect[:3])
            for p in range(3):
                ys, xs = slice(rect[3 + p], rect[3 + p] + rect[7]), slice(rect[p], rect[p] + rect[6])
                sl = slice(p * C, (p + 1) * C)
                assert torch.equal(dx[sl, ys, xs], ref[lvl][0][sl, ys, xs]), (wave, trial, lvl, p)
                assert torch.equal(dyh[sl, :, ys, xs], ref[lvl][1][sl, :, ys, xs]), (wave, trial, lvl, p)
                mask = torch.ones(n, n, dtype=torch.bool, device=cuda)
                mask[ys, xs] = False
                assert bool((dx[sl][:, mask] == 123.0).all()) and bool((dyh[sl][:, :, mask] == 123.0).all())
                assert float(ref[lvl][0][sl][:, mask].abs().sum()) == 0, (wave, trial, lvl, p, rect, win)
                assert float(ref[lvl][1][sl][:, :, mask].abs().sum()) == 0, (wave, trial, lvl, p, rect, win)
            win, strided, src = rect + [C, 0], 1, dx
