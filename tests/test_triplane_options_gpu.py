"""TriPlaneVolume options outside the README configurations against the reference class
itself (tests/golden/triplane_options_reference.npz, generator make_golden_options.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

C, R, SCALE, WAVE = 2, 32, 4, "bior2.2"


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "triplane_options_reference.npz"))


def _vol(dev, g, tag, **kw):
    from trinerflet_amd.triplaneencoder.triplane_encoder import TriPlaneVolume
    vol = TriPlaneVolume(number_of_features=C, plane_resolution=R, inner_multi_res_scale=SCALE, wavelet_type=WAVE,
                         lbound=float(g["bound"]), plane_dtype=torch.float32, **kw).to(dev)
    with torch.no_grad():
        vol.planes_features.copy_(torch.from_numpy(g[f"{tag}/ll"]))
        for i, p in enumerate(vol.planes_features_wavelet_coefs):
            p.copy_(torch.from_numpy(g[f"{tag}/coef{i}"]))
    return vol


def _close(a, b, tol=2e-5):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    scale = max(1.0, float(np.abs(b).max()))
    assert a.shape == b.shape, (a.shape, b.shape)
    assert float(np.abs(a - b).max()) <= tol * scale, float(np.abs(a - b).max())


def test_tanh_on_features(cuda, g):
    vol = _vol(cuda, g, "tanh", apply_activation_on_features=True)
    assert not vol.is_plain()
    _close(vol.get_planes(), g["tanh/planes"])
    _close(vol(torch.from_numpy(g["xyz"]).to(cuda), float(g["bound"])), g["tanh/forward"])


def test_lbound_auto_scale(cuda, g):
    vol = _vol(cuda, g, "lbound", lbound_auto_scale=True)
    assert [n for n, _ in vol.named_parameters()] == list(g["lbound/param_names"])
    with torch.no_grad():
        vol.lbound_scale.copy_(torch.from_numpy(g["lbound/scale"]))
    _close(vol.get_lbound_scale(), g["lbound/get_lbound_scale"])
    f = vol(torch.from_numpy(g["xyz"]).to(cuda), float(g["bound"]))
    _close(f, g["lbound/forward"])
    (gs,) = torch.autograd.grad(f, [vol.lbound_scale], torch.from_numpy(g["cot"]).to(cuda), retain_graph=True)
    _close(gs, g["lbound/g_scale"], tol=2e-4)
    groups = vol.get_params2(0.01)
    assert [gr["lr"] for gr in groups] == list(g["lbound/params2_lrs"])
    assert [len(gr["params"]) for gr in groups] == list(g["lbound/params2_sizes"])
    # the rest of the chain (planes, coefficients) is differentiable through the IDWT kernels
    f.sum().backward()
    assert vol.planes_features.grad is not None and float(vol.planes_features.grad.abs().sum()) > 0


def test_learned_rotation(cuda, g):
    vol = _vol(cuda, g, "rot", learn_rotation_axis=True)
    with torch.no_grad():
        vol.rotation_matrix.copy_(torch.from_numpy(g["rot/rotation_matrix"]))
    f = vol(torch.from_numpy(g["xyz"]).to(cuda), float(g["bound"]))
    _close(f, g["rot/forward"], tol=1e-4)
    (gr,) = torch.autograd.grad(f, [vol.rotation_matrix], torch.from_numpy(g["cot"]).to(cuda))
    _close(gr, g["rot/g_rot"], tol=2e-3)


def test_nested_zoom_planes(cuda, g):
    vol = _vol(cuda, g, "up", upscale_ratio_bound=0.5, upscale_levels=2)
    assert vol.upscale_enabled and len(vol.get_wavelet_features_upscaled()) == int(g["up/n_upscaled_features"])
    assert vol.upscale_base_resolution_lst == list(g["up/base_resolution"])
    assert vol.upscale_base_corner_lst == list(g["up/base_corner"])
    np.testing.assert_allclose(vol.upscale_bound_ratio_lst, g["up/bound_ratio"])
    with torch.no_grad():
        for i, p in enumerate(vol.upscale_wavelet_lst):
            p.copy_(torch.from_numpy(g[f"up/wavelet{i}"]))
    planes = vol.get_planes()
    assert isinstance(planes, list) and len(planes) == 3
    for i, p in enumerate(planes):
        _close(p, g[f"up/planes{i}"])
    f = vol(torch.from_numpy(g["xyz"]).to(cuda), float(g["bound"]))
    _close(f, g["up/forward"])
    f.sum().backward()                       # gradients reach the nested wavelets and the base coefficients
    assert all(float(p.grad.abs().sum()) > 0 for p in vol.upscale_wavelet_lst)
    assert float(vol.planes_features_wavelet_coefs[0].grad.abs().sum()) > 0


def test_partially_learnable_levels_and_partial_builds(cuda, g):
    vol = _vol(cuda, g, "cur", inner_multi_res_scale_current=2)
    assert len(vol.planes_features_wavelet_coefs) == int(g["cur/n_learnable"]) == 1
    _close(vol.get_planes(), g["cur/planes"])
    vol = _vol(cuda, g, "partial")
    for tag, kw in (("max_res16", dict(max_res=16)), ("max_scale2", dict(max_scale=2)), ("full", dict())):
        vol.reset_cahce()
        _close(vol.get_planes(**kw), g[f"partial/{tag}"])
    vol.reset_cahce()
    allres = vol.get_planes(get_all_resolutions=True)
    assert len(allres) == int(g["partial/all_n"])
    for i, a in enumerate(allres):
        _close(a, g[f"partial/all{i}"])
    vol.reset_cahce()
    lb, feats, grid = vol.get_grid_features(4)
    assert lb == float(g["grid/lbound"])
    _close(grid, g["grid/grid"])
    _close(feats, g["grid/features"])


@pytest.mark.parametrize("tag,wave", [("wbr22", "bior2.2"), ("wbr68", "bior6.8")])
def test_wavelet_base_resolution(cuda, g, tag, wave):
    """Levels below wavelet_base_resolution are synthesised without the zero halo (triplane_encoder.py:391-393) and have
    the uncropped analysis sizes (:190-196): level sizes, planes and lookup vs the reference class."""
    from trinerflet_amd.triplaneencoder.triplane_encoder import TriPlaneVolume
    res, sc, wbr = (int(v) for v in g[f"{tag}/cfg"])
    vol = TriPlaneVolume(number_of_features=C, plane_resolution=res, inner_multi_res_scale=sc, wavelet_type=wave,
                         lbound=float(g["bound"]), plane_dtype=torch.float32, wavelet_base_resolution=wbr).to(cuda)
    shapes = [p.shape[-1] for p in vol.planes_features_wavelet_coefs] + [vol.planes_features.shape[-1]]
    assert shapes == list(g[f"{tag}/shapes"])
    with torch.no_grad():
        vol.planes_features.copy_(torch.from_numpy(g[f"{tag}/ll"]))
        for i, p in enumerate(vol.planes_features_wavelet_coefs):
            p.copy_(torch.from_numpy(g[f"{tag}/coef{i}"]))
    _close(vol.get_planes(), g[f"{tag}/planes"])
    f = vol(torch.from_numpy(g["xyz"]).to(cuda), float(g["bound"]))
    _close(f, g[f"{tag}/forward"])
    f.sum().backward()            # the crop is differentiable: every level receives a gradient
    assert all(float(p.grad.abs().sum()) > 0 for p in vol.planes_features_wavelet_coefs)


def test_options_switch_the_fused_path_off_and_still_render(cuda):
    from trinerflet_amd.nerf.network import NeRFNetwork
    m = NeRFNetwork(encoding="triplane_wavelet", bound=1.0, cuda_ray=True, hidden_dim=64, hidden_dim_color=64,
                    triplane_channels=16, triplane_resolution=64, triplane_wavelet_levels=2, wavelet_type="bior2.2",
                    lbound_auto_scale=True, upscale_ratio_bound=0.5, upscale_levels=1).to(cuda)
    assert not m._fused_ok()
    x = torch.rand(500, 3, device=cuda) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(500, 3, device=cuda), dim=-1)
    sigma, rgb = m(x, d)
    assert sigma.shape == (500,) and rgb.shape == (500, 3) and torch.isfinite(rgb).all()
    (sigma.sum() + rgb.sum()).backward()
    assert m.encoder.lbound_scale.grad is not None and m.encoder.upscale_wavelet_lst[0].grad is not None
