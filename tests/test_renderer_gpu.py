"""GPU tests of the renderer-level glue: density-grid upkeep, the inference loop of
run_cuda against its training branch, and the non-cuda_ray renderer."""
import numpy as np
import pytest
import torch

from oracle import cref
from trinerflet_amd import synthetic

pytestmark = pytest.mark.gpu


def _model(dev, cuda_ray=True):
    from trinerflet_amd.nerf.network import NeRFNetwork
    m = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=cuda_ray, density_thresh=10, hidden_dim=64,
                    hidden_dim_color=64, triplane_channels=16, triplane_resolution=64, triplane_wavelet_levels=4,
                    wavelet_type="bior6.8").to(dev)
    synthetic.init_field_parameters(m, seed=5)
    return m


def test_update_extra_state_full_refresh(cuda):
    """Full refresh (renderer.py:459-489) with a density that is constant inside each grid cell, so the random
    in-cell jitter cannot change the result: grid, EMA, mean, threshold and bitfield must equal the restatement."""
    m = _model(cuda)
    H, cas = m.grid_size, m.cascade

    def cell_density(x):  # x: [n,3] world positions; which cascade is encoded by the bound of the caller
        return x

    state = {"cas": 0}
    bounds = [min(2 ** c, m.bound) for c in range(cas)]

    def fake_density(xyz):
        b = bounds[state["cas"]]
        state["cas"] = (state["cas"] + 1) % cas
        hg = b / H
        cell = torch.clamp(((xyz / (b - hg) + 1) * (H - 1) / 2).round(), 0, H - 1)   # inverse of the cell centre map
        val = 20.0 * torch.exp(-((cell - (H - 1) / 2) ** 2).sum(-1) / (2 * (0.15 * H) ** 2)) * (1 + 0.5 * (b > 1))
        return {"sigma": val, "geo_feat": None}
    m.density = fake_density
    m.density_grid.zero_()
    m.density_grid[0, :1000] = -1       # untrained cells must stay untouched (valid_mask, :525)
    m.update_extra_state()
    ax = np.arange(H)
    cells = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    mort = cref.morton3D(cells.astype(np.int32))
    expect = np.zeros((cas, H ** 3), np.float32)
    for c, b in enumerate(bounds):
        v = 20.0 * np.exp(-((cells - (H - 1) / 2) ** 2).sum(-1) / (2 * (0.15 * H) ** 2)) * (1 + 0.5 * (b > 1))
        expect[c, mort] = v                           # max(0 * 0.95, v)
    expect[0, :1000] = -1
    got = m.density_grid.cpu().numpy()
    # the jitter reaches exactly +-half a cell, so a few samples on a cell boundary may round to the neighbour
    bad = ~np.isclose(got, expect, rtol=1e-5, atol=1e-6)
    assert bad.sum() < 200, bad.sum()
    mean = np.clip(got, 0, None).mean()
    assert abs(m.mean_density - mean) < 1e-5 * mean and m.iter_density == 1
    assert abs(mean - np.clip(expect, 0, None).mean()) < 1e-4 * mean
    thresh = min(m.mean_density, m.density_thresh)
    assert np.array_equal(m.density_bitfield.cpu().numpy(), cref.packbits(got, thresh))
    # second refresh: EMA max(grid * 0.95, new) leaves equal values unchanged
    m.update_extra_state()
    bad2 = ~np.isclose(m.density_grid.cpu().numpy(), expect, rtol=1e-5, atol=1e-6)
    assert bad2.sum() < 400
    # partial refresh path (iter_density >= 16) runs and keeps the invariants
    m.iter_density = 16
    m.update_extra_state()
    g3 = m.density_grid.cpu().numpy()
    assert np.all(g3[0, :1000] == -1) and np.all(g3[expect >= 0] >= expect[expect >= 0] * 0.95 - 1e-6)


def test_mark_untrained_grid(cuda):
    m = _model(cuda)
    poses = synthetic.hemisphere_poses(6, seed=2)
    fl = 800 / (2 * np.tan(0.6911 / 2))
    intr = (fl, fl, 400.0, 400.0)
    m.density_grid.zero_()
    m.mark_untrained_grid(poses, intr)
    got = m.density_grid.cpu().numpy()
    # brute-force restatement of renderer.py:383-446 on a random subset of cells
    rng = np.random.default_rng(0)
    H = m.grid_size
    cells = rng.integers(0, H, (4000, 3))
    mort = cref.morton3D(cells.astype(np.int32))
    for cas in range(m.cascade):
        b = min(2 ** cas, m.bound)
        hg = b / H
        w = (2 * cells / (H - 1) - 1) * (b - hg)
        seen = np.zeros(len(cells), bool)
        for P in poses:
            cam = (w - P[:3, 3]) @ P[:3, :3]
            seen |= (cam[:, 2] > 0) & (np.abs(cam[:, 0]) < 400 / fl * cam[:, 2] + hg * 2) & \
                    (np.abs(cam[:, 1]) < 400 / fl * cam[:, 2] + hg * 2)
        assert np.array_equal(got[cas, mort] == -1, ~seen)


def test_eval_render_matches_train_branch(cuda):
    """run_cuda's inference loop (march_rays / composite_rays / compaction) vs its training branch on the same rays
    without perturbation: same image, alpha and normalised depth up to the different T bookkeeping."""
    m = _model(cuda)
    bf = synthetic.sphere_bitfield(128, 2, 1.5, 0.8, 0.5)
    m.density_bitfield.copy_(torch.from_numpy(bf).to(cuda))
    with torch.no_grad():  # make the medium absorbing enough for early termination to occur
        m.sigma_net[1].weight[0].add_(0.6)
    o, d = synthetic.training_rays(2048, n_cams=4, seed=9)
    ro, rd = torch.from_numpy(o).to(cuda)[None], torch.from_numpy(d).to(cuda)[None]
    m.train()
    m.mean_count = 0
    with torch.no_grad():
        tr = m.render(ro, rd, staged=False, bg_color=0, perturb=False, force_all_rays=True, T_thresh=1e-4)
    m.eval()
    with torch.no_grad():
        ev = m.render(ro, rd, staged=True, bg_color=0, perturb=False, T_thresh=1e-4)
    np.testing.assert_allclose(ev["image"].cpu().numpy(), tr["image"].cpu().numpy(), rtol=0, atol=2e-3)
    np.testing.assert_allclose(ev["weights_sum"].cpu().numpy().reshape(-1), tr["weights_sum"].cpu().numpy(), rtol=0, atol=2e-3)
    assert float(ev["weights_sum"].max()) > 0.5


def test_non_cuda_ray_renderer_runs(cuda):
    """NeRFRenderer.run (renderer.py:126-254) on the modular density()/color() path."""
    m = _model(cuda, cuda_ray=False)
    o, d = synthetic.training_rays(256, n_cams=2, seed=1)
    with torch.no_grad():
        out = m.render(torch.from_numpy(o).to(cuda)[None], torch.from_numpy(d).to(cuda)[None], staged=True,
                       max_ray_batch=100, num_steps=64, upsample_steps=0, bg_color=1)
    assert out["image"].shape == (1, 256, 3) and torch.isfinite(out["image"]).all()
    assert float(out["image"].min()) >= 0 and float(out["image"].max()) <= 1.0 + 1e-4
    # with the importance resampling pass (renderer.py:176-213), eval = deterministic strata, train = random + grads
    m.eval()
    with torch.no_grad():
        up = m.render(torch.from_numpy(o).to(cuda)[None], torch.from_numpy(d).to(cuda)[None], staged=True,
                      max_ray_batch=100, num_steps=32, upsample_steps=32, bg_color=1)
        up2 = m.render(torch.from_numpy(o).to(cuda)[None], torch.from_numpy(d).to(cuda)[None], staged=True,
                       max_ray_batch=100, num_steps=32, upsample_steps=32, bg_color=1)
    assert torch.isfinite(up["image"]).all() and torch.equal(up["image"], up2["image"])
    m.train()
    res = m.run(torch.from_numpy(o[:64]).to(cuda), torch.from_numpy(d[:64]).to(cuda), num_steps=16, upsample_steps=16,
                bg_color=1, perturb=True)
    res["image"].sum().backward()
    assert m.sigma_net[0].weight.grad is not None and res["weights_sum"].shape == (64,)


def test_device_driven_inference_loop_equals_host_driven(cuda):
    """The alive-ray loop with its sizes on the device (tnl_infer_plan / *_dev) renders exactly what the
    host-driven loop (one survivor-count read-back per iteration, renderer.py:345,364) renders; the host-driven loop
    is the one tied to the oracle in tests/test_raymarching_gpu.py::test_inference_loop."""
    from trinerflet_amd import synthetic
    m = _model(cuda)
    synthetic.init_field_parameters(m, seed=5)
    with torch.no_grad():
        m.sigma_net[1].weight[0] += 0.6           # some opacity, so that rays terminate at different iterations
    m.density_bitfield.copy_(torch.from_numpy(synthetic.sphere_bitfield(128, m.cascade, float(m.bound), 0.8, 0.2)).to(cuda))
    m.eval()
    o, d = synthetic.training_rays(5000, n_cams=6, seed=4)
    o, d = torch.from_numpy(o).to(cuda)[None], torch.from_numpy(d).to(cuda)[None]
    outs = []
    with torch.no_grad():
        for dev_loop in (False, True):
            for max_steps in (64, 1024):
                outs.append(m.render(o, d, staged=True, bg_color=0.5, perturb=False, max_steps=max_steps,
                                     device_loop=dev_loop))
    for a, b in ((outs[0], outs[2]), (outs[1], outs[3])):
        for k in ("image", "depth", "weights_sum"):
            # rays that miss the box carry depth = 0 / 0 = nan in the reference as well (renderer.py:370)
            assert torch.equal(torch.nan_to_num(a[k], nan=-1.0), torch.nan_to_num(b[k], nan=-1.0)), k
    assert float(outs[1]["weights_sum"].max()) > 0.5 and not torch.equal(outs[0]["image"], outs[1]["image"])
    # wider iterations (infer_min_step=8): with a cap no ray reaches (4096) the images are identical bit for bit
    with torch.no_grad():
        ref = m.render(o, d, staged=True, bg_color=0.5, perturb=False, max_steps=4096, device_loop=True)
        wide = m.render(o, d, staged=True, bg_color=0.5, perturb=False, max_steps=4096, infer_min_step=8)
    for k in ("image", "depth", "weights_sum"):
        assert torch.equal(torch.nan_to_num(ref[k], nan=-1.0), torch.nan_to_num(wide[k], nan=-1.0)), k


def test_planes_get_gradient_right_after_a_grid_refresh(cuda):
    """update_extra_state queries the field under no_grad; the texel-major plane copy it caches must not be handed to
    the differentiable lookup of the same step (the reference refreshes the grid between get_planes() and
    train_step, utils.py:1138-1147)."""
    from trinerflet_amd import synthetic
    m = _model(cuda)
    m.train()
    m.encoder.reset_cahce()
    m.encoder.get_planes()
    m.update_extra_state()
    o, d = synthetic.training_rays(256, n_cams=2, seed=3)
    out = m.render(torch.from_numpy(o).to(cuda)[None], torch.from_numpy(d).to(cuda)[None], staged=False, bg_color=0,
                   perturb=True, force_all_rays=True)
    out["image"].sum().backward()
    g = m.encoder.planes_features.grad
    assert g is not None and float(g.abs().sum()) > 0
    assert float(m.encoder.planes_features_wavelet_coefs[0].grad.abs().sum()) > 0


def test_partial_refresh_picks_occupied_cells_without_a_read_back(cuda):
    """Partial refresh (renderer.py:494-520): H^3/4 uniform cells + H^3/4 draws from the currently occupied cells per
    cascade.  The occupied draws are made from the running count of occupied cells (no nonzero() read-back): every
    occupied cell of a small occupied set is hit, nothing but uniform draws happens when no cell is occupied."""
    m = _model(cuda)
    H, cas = m.grid_size, m.cascade
    m.density = lambda xyz: {"sigma": torch.full((xyz.shape[0],), 7.0, device=xyz.device), "geo_feat": None}
    g = torch.Generator(device="cpu").manual_seed(5)
    occ = torch.randperm(H ** 3, generator=g)[:1000].to(cuda)
    m.density_grid.zero_()
    m.density_grid[:, occ] = 1.0
    m.iter_density = 16
    torch.manual_seed(1)
    m.update_extra_state()
    grid = m.density_grid
    assert bool((grid[:, occ] == 7.0).all())                       # each of the 1000 occupied cells was drawn (524 288 draws)
    touched = (grid == 7.0).sum(1).cpu().numpy()
    N = H ** 3 // 4
    assert (touched >= 1000).all() and (touched <= N + 1000).all() and (touched > 0.85 * N).all()
    assert bool(((grid == 7.0) | (grid == 0.0)).all())             # untouched cells keep max(0 * 0.95, -1 -> invalid) = 0
    # nothing occupied: only the uniform draws, and no out-of-range write
    m.density_grid.zero_()
    m.iter_density = 16
    m.update_extra_state()
    touched = (m.density_grid == 7.0).sum(1).cpu().numpy()
    assert (touched <= N).all() and (touched > 0.85 * N).all()
