"""GPU parity of the wavelet-triplane kernels (IDWT levels, adjoint, layout change, lookup) against the
golden vectors produced by the reference + PyWavelets, and against the CPU oracle on seeded inputs."""
import os

import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu


def _t(a, dev, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dtype)


def _vol(dev, C, R, scale, wave, plane_dtype=torch.float32):
    from trinerflet_amd.triplaneencoder.triplane_encoder import TriPlaneVolume
    return TriPlaneVolume(number_of_features=C, plane_resolution=R, inner_multi_res_scale=scale, wavelet_type=wave,
                          plane_dtype=plane_dtype).to(dev)


@pytest.mark.parametrize("wave", cref.WAVELETS)
def test_idwt_matches_reference_golden(cuda, golden_dir, wave):
    """tests/golden/triplane_reference.npz: TriPlaneVolume.get_planes() of the REFERENCE class (float64, pywt)."""
    g = np.load(os.path.join(golden_dir, "triplane_reference.npz"))
    vol = _vol(cuda, 2, 32, 4, wave)
    with torch.no_grad():
        vol.planes_features.copy_(_t(g[f"idwt/{wave}/ll"], cuda))
        vol.planes_features_wavelet_coefs[0].copy_(_t(g[f"idwt/{wave}/coef0"], cuda))
        vol.planes_features_wavelet_coefs[1].copy_(_t(g[f"idwt/{wave}/coef1"], cuda))
    planes = vol.get_planes()
    ref = g[f"idwt/{wave}/planes"]
    assert planes.shape == ref.shape
    # fp32 kernel with fp32 taps vs float64 reference: tolerance = a few fp32 ulps of the accumulated magnitude
    np.testing.assert_allclose(planes.detach().cpu().numpy(), ref, rtol=0, atol=3e-6 * np.abs(ref).max())


@pytest.mark.parametrize("wave", cref.WAVELETS)
def test_idwt_matches_pywt_golden(cuda, golden_dir, wave):
    g = np.load(os.path.join(golden_dir, "idwt_pywt.npz"))
    from trinerflet_amd.triplaneencoder.triplane_encoder import _IDWTLevel, WAVELET_IDS
    x = _t(g[f"{wave}/ll"], cuda).view(1, 2, 8, 8)
    for lvl in range(2):
        n = x.shape[-1]
        x = _IDWTLevel.apply(x, _t(g[f"{wave}/yh{lvl}"], cuda).view(1, 2, 3, n, n), WAVELET_IDS[wave])
    ref = g[f"{wave}/planes"]
    np.testing.assert_allclose(x.view(2, 32, 32).cpu().numpy(), ref, rtol=0, atol=3e-6 * np.abs(ref).max())


@pytest.mark.parametrize("wave,C,R,scale", [("bior6.8", 3, 256, 4), ("haar", 2, 128, 8), ("bior4.4", 1, 192, 2),
                                            ("bior2.6", 2, 72, 2), ("bior2.2", 2, 40, 4)])
def test_build_planes_and_adjoint_vs_oracle(cuda, wave, C, R, scale):
    """Seeded random coefficients at sizes that exercise multi-tile grids, ragged tiles and all wavelets;
    forward vs the C oracle; backward (autograd) vs the oracle adjoint; plus the <Ax,y> = <x,A^T y> identity."""
    torch.manual_seed(0)
    vol = _vol(cuda, C, R, scale, wave)
    with torch.no_grad():
        vol.planes_features.normal_(0, 0.5)
        for p in vol.planes_features_wavelet_coefs:
            p.normal_(0, 0.3)
    planes = vol.get_planes()
    ll = vol.planes_features.detach().cpu().numpy()
    coefs = [p.detach().cpu().numpy() for p in vol.planes_features_wavelet_coefs]
    ref = cref.build_planes(ll, coefs, wave)
    scale_ = np.abs(ref).max()
    np.testing.assert_allclose(planes.detach().cpu().numpy(), ref, rtol=0, atol=3e-6 * scale_)
    cot = torch.randn_like(planes)
    planes.backward(cot)
    dll, dcoefs = cref.build_planes_adj(cot.cpu().numpy(), len(coefs), wave)
    np.testing.assert_allclose(vol.planes_features.grad.cpu().numpy(), dll, rtol=0, atol=5e-6 * np.abs(dll).max())
    for p, d in zip(vol.planes_features_wavelet_coefs, dcoefs):
        np.testing.assert_allclose(p.grad.cpu().numpy(), d, rtol=0, atol=5e-6 * np.abs(d).max())
    # adjoint identity in float64 accumulate
    lhs = (planes.detach().double() * cot.double()).sum().item()
    rhs = (vol.planes_features.detach().double() * vol.planes_features.grad.double()).sum().item()
    rhs += sum((p.detach().double() * p.grad.double()).sum().item() for p in vol.planes_features_wavelet_coefs)
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


def test_layout_roundtrip(cuda):
    from trinerflet_amd.triplaneencoder.triplane_encoder import _ToTexelMajor
    torch.manual_seed(1)
    for C, R in ((16, 96), (5, 70)):
        cm = torch.randn(3, C, R, R, device=cuda, requires_grad=True)
        tm = _ToTexelMajor.apply(cm, False)
        assert torch.equal(tm, cm.detach().permute(0, 2, 3, 1).contiguous())
        th = _ToTexelMajor.apply(cm, True)
        assert torch.equal(th, cm.detach().permute(0, 2, 3, 1).contiguous().half())
        g = torch.randn_like(tm)
        tm.backward(g)
        assert torch.equal(cm.grad, g.permute(0, 3, 1, 2).contiguous())


def test_sample_matches_reference_golden(cuda, golden_dir):
    """tests/golden/triplane_reference.npz sample/*: the REFERENCE TriPlaneVolume.forward + its autograd VJP."""
    from trinerflet_amd.triplaneencoder.triplane_encoder import TriPlaneVolume
    g = np.load(os.path.join(golden_dir, "triplane_reference.npz"))
    vol = TriPlaneVolume(number_of_features=4, plane_resolution=32, inner_multi_res_scale=4, wavelet_type="haar",
                         plane_dtype=torch.float32).to(cuda)
    planes = _t(g["sample/planes"], cuda).requires_grad_(True)
    xyz = _t(g["sample/xyz"], cuda)
    bound = float(g["sample/bound"])
    feats = vol.sample_from_planes(xyz, plane_features=planes, lbound=bound).view(xyz.shape[0], -1)
    np.testing.assert_allclose(feats.detach().cpu().numpy(), g["sample/feats"], rtol=0, atol=2e-6)
    feats.backward(_t(g["sample/cot"], cuda))
    np.testing.assert_allclose(planes.grad.cpu().numpy(), g["sample/dplanes"], rtol=0, atol=1e-5)
    # forward() through the cache, fp16 storage (fast mode): within fp16 rounding of the texels
    vol16 = TriPlaneVolume(number_of_features=4, plane_resolution=32, inner_multi_res_scale=4, wavelet_type="haar").to(cuda)
    vol16.last_used_planes = planes.detach()
    out16 = vol16(xyz, bound)
    np.testing.assert_allclose(out16.cpu().numpy(), g["sample/feats"], rtol=0, atol=1.5e-3)


def test_sample_vs_oracle_large(cuda):
    from trinerflet_amd.triplaneencoder.triplane_encoder import _ToTexelMajor, _Sample
    rng = np.random.default_rng(3)
    C, R, N, bound = 16, 128, 20000, 1.5
    planes = rng.standard_normal((3, C, R, R)).astype(np.float32)
    xyz = ((rng.random((N, 3)) * 2 - 1) * bound * 1.05).astype(np.float32)
    tm = _ToTexelMajor.apply(_t(planes, cuda), False).requires_grad_(True)
    feats = _Sample.apply(tm, _t(xyz, cuda), bound)
    np.testing.assert_allclose(feats.detach().cpu().numpy(), cref.triplane_sample(planes, xyz, bound), rtol=0, atol=3e-6)
    cot = rng.standard_normal((N, 3 * C)).astype(np.float32)
    feats.backward(_t(cot, cuda))
    ref = cref.triplane_sample_bwd(cot, xyz, bound, C, R)
    got = tm.grad.permute(0, 3, 1, 2).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5 * np.abs(ref).max())


def test_half_finest_level_and_layout(cuda):
    """TrainStep's fast path: finest level written as fp16 + fp16 layout change == fp32 level, rounded."""
    from trinerflet_amd.triplaneencoder.triplane_encoder import (_IDWTLevel, WAVELET_IDS, half_to_texel_major,
                                                                  idwt_level_half)
    torch.manual_seed(2)
    for wave, C, n in (("bior6.8", 16, 128), ("bior4.4", 8, 40), ("haar", 8, 64)):
        x = torch.randn(3, C, n, n, device=cuda) * 0.3
        yh = torch.randn(3, C, 3, n, n, device=cuda) * 0.2
        ref = _IDWTLevel.apply(x, yh, WAVELET_IDS[wave])
        h = idwt_level_half(x, yh, WAVELET_IDS[wave])
        # identical up to fp32 contraction differences between the two template instantiations: a handful of
        # exact rounding ties may land on the neighbouring fp16 value
        d = (h.float() - ref.half().float()).abs()
        assert h.dtype == torch.float16 and float((d > 0).float().mean()) < 1e-4 and float(d.max()) <= 2 ** -10
        tm = half_to_texel_major(h)
        assert torch.equal(tm, h.permute(0, 2, 3, 1).contiguous())
