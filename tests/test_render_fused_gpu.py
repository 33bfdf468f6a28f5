"""The inference render as one persistent kernel (csrc/render.hip, `render(..., fused_render=True)`) against the
alive-ray loop it replaces (renderer.py:324-374; device-driven loop, itself bit-identical to the host-driven one and
pinned to the reference's eval branch by tests/test_reference_fixtures_gpu.py): same samples per ray in the same order,
so image / weights / depth agree to fp32 rounding; also on the reference fixture's rays against the reference-run image."""
import os

import numpy as np
import pytest
import torch

from trinerflet_amd import synthetic

pytestmark = pytest.mark.gpu


def _model(dev, C=16, R=128, H=64, seed=5):
    from trinerflet_amd.nerf.network import NeRFNetwork
    m = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, density_thresh=10, hidden_dim=H,
                    hidden_dim_color=H, triplane_channels=C, triplane_resolution=R, triplane_wavelet_levels=4,
                    wavelet_type="bior6.8").to(dev)
    synthetic.init_field_parameters(m, seed=seed)
    m.density_bitfield.copy_(torch.from_numpy(synthetic.sphere_bitfield(128, 2, 1.5, 0.8, 0.3)).to(dev))
    m.eval()
    # density_scale for a median optical depth of ~12 through the shell: rays end by transmittance as well as by leaving
    g = torch.Generator().manual_seed(3)
    p = torch.randn(8000, 3, generator=g)
    p = (p / p.norm(dim=-1, keepdim=True) * (0.3 + 0.5 * torch.rand(8000, 1, generator=g))).to(dev)
    with torch.no_grad():
        med = float(m.density(p)["sigma"].float().median())
    m.density_scale = float(np.float32(12.0 / med))
    return m


@pytest.mark.parametrize("C,H", [(16, 64), (32, 64), (48, 128)])
@pytest.mark.parametrize("max_steps", [1024, 4096])
def test_fused_render_equals_the_loop(cuda, C, H, max_steps):
    m = _model(cuda, C=C, H=H)
    o, d = synthetic.training_rays(6000, n_cams=6, seed=3)
    ro, rd = torch.from_numpy(o).to(cuda)[None], torch.from_numpy(d).to(cuda)[None]
    with torch.no_grad():
        loop = m.render(ro, rd, staged=True, bg_color=0.3, perturb=False, max_steps=max_steps, T_thresh=1e-4,
                        device_loop=True)
        one = m.render(ro, rd, staged=True, bg_color=0.3, perturb=False, max_steps=max_steps, T_thresh=1e-4,
                       fused_render=True)
    ws_l, ws_o = loop["weights_sum"].reshape(-1).cpu().numpy(), one["weights_sum"].reshape(-1).cpu().numpy()
    img_l, img_o = loop["image"][0].cpu().numpy(), one["image"][0].cpu().numpy()
    assert 0.05 < (ws_l > 0.5).mean() < 0.6 and (ws_l > 1 - 2e-4).sum() > 20      # hits, misses, T-terminated rays
    np.testing.assert_allclose(ws_o, ws_l, rtol=0, atol=2e-6)
    np.testing.assert_allclose(img_o, img_l, rtol=0, atol=2e-6)
    dl, do = loop["depth"][0].cpu().numpy(), one["depth"][0].cpu().numpy()
    hit = np.isfinite(dl)
    np.testing.assert_allclose(do[hit], dl[hit], rtol=0, atol=5e-6)
    assert np.isnan(do[~hit]).all()
    # most rays agree to the last bit (same samples, same order, same arithmetic)
    assert (ws_o == ws_l).mean() > 0.95, float((ws_o == ws_l).mean())


def test_fused_render_perturbed_and_empty_and_scaled(cuda):
    """Perturbation (the first iteration's noise), density_scale != 1, an empty occupancy grid, zero rays."""
    m = _model(cuda)
    m.density_scale = m.density_scale * 0.37
    o, d = synthetic.training_rays(3000, n_cams=4, seed=8)
    ro, rd = torch.from_numpy(o).to(cuda)[None], torch.from_numpy(d).to(cuda)[None]
    with torch.no_grad():
        torch.manual_seed(11)
        loop = m.render(ro, rd, staged=True, bg_color=1.0, perturb=True, max_steps=1024, device_loop=True)
        torch.manual_seed(11)
        one = m.render(ro, rd, staged=True, bg_color=1.0, perturb=True, max_steps=1024, fused_render=True)
    np.testing.assert_allclose(one["image"].cpu().numpy(), loop["image"].cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(one["weights_sum"].cpu().numpy(), loop["weights_sum"].cpu().numpy(), rtol=0, atol=2e-6)
    m.density_bitfield.zero_()
    with torch.no_grad():
        e = m.render(ro, rd, staged=True, bg_color=1.0, perturb=False, fused_render=True)
    assert float(e["weights_sum"].abs().max()) == 0.0 and float((e["image"] - 1.0).abs().max()) == 0.0
    with torch.no_grad():
        z = m.render(ro[:, :0], rd[:, :0], staged=True, bg_color=1.0, perturb=False, fused_render=True)
    assert z["image"].shape == (1, 0, 3)


def test_fused_render_matches_reference_eval_branch(cuda, golden_dir):
    """F-INFER: the reference's run_cuda eval branch (renderer.py:324-374, run in the build container over the oracle's
    kernels, tests/golden/network_reference.npz) against the one-kernel render."""
    from tests.test_reference_fixtures_gpu import _cfg, _model as ref_model
    ref = np.load(os.path.join(golden_dir, "network_reference.npz"))
    C, R, scale, H, N, max_steps, bound, lam, bg, lr, min_near = _cfg(ref)
    m = ref_model(ref, cuda)
    m.eval()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    with torch.no_grad():
        out = m.render(t(ref["rays/o"])[None], t(ref["rays/d"])[None], staged=True, bg_color=bg, perturb=False,
                       dt_gamma=0, max_steps=max_steps, T_thresh=1e-4, fused_render=True)
    hit = np.isfinite(ref["infer/depth"])
    np.testing.assert_allclose(out["image"][0].cpu().numpy(), ref["infer/image"], atol=2e-3)
    np.testing.assert_allclose(out["weights_sum"].reshape(-1).cpu().numpy(), ref["infer/weights_sum"], atol=2e-3)
    np.testing.assert_allclose(out["depth"][0].cpu().numpy()[hit], ref["infer/depth"][hit], atol=2e-3)


@pytest.mark.parametrize("C,H", [(16, 64), (48, 128)])
def test_marching_in_bounded_pieces_changes_no_bit(cuda, C, H):
    """tnl_render_work (round 6): a ray spends a bounded amount of marching work per trip and rides along without a sample
    until it reaches its next one, instead of stalling the wave's other rays for its whole walk through empty cells.  The
    probes, adds and comparisons are the unbounded march's in the same order: images, weights and depths for every budget --
    down to one probe per trip -- are the unbounded render's to the bit (max_steps 4096: the `--test` setting, where an
    empty cell is 28-55 chain steps; a hollow shell, so that rays also skip INSIDE the object)."""
    from trinerflet_amd import _lib as L
    m = _model(cuda, C=C, H=H)
    o, d = synthetic.training_rays(20000, n_cams=6, seed=4)
    ro, rd = torch.from_numpy(o).to(cuda)[None], torch.from_numpy(d).to(cuda)[None]
    prev = L.lib().tnl_render_work(L.i32(-1))
    assert prev == 96                                              # the default in force
    outs = {}
    try:
        for work in (0, 8, 31, 96, 4096):
            L.lib().tnl_render_work(L.i32(work))
            assert L.lib().tnl_render_work(L.i32(-1)) == work
            with torch.no_grad():
                outs[work] = m.render(ro, rd, staged=True, bg_color=0.3, perturb=False, max_steps=4096, T_thresh=1e-4)
    finally:
        L.lib().tnl_render_work(L.i32(prev))
    ws = outs[0]["weights_sum"].reshape(-1)
    assert 0.05 < float((ws > 0.5).float().mean()) < 0.6
    for work, out in outs.items():
        assert torch.equal(out["image"], outs[0]["image"]) and torch.equal(out["weights_sum"], outs[0]["weights_sum"]), work
        assert torch.equal(torch.nan_to_num(out["depth"]), torch.nan_to_num(outs[0]["depth"])), work
