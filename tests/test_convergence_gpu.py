"""End-to-end sanity of the training dynamics (fused step, GradScaler, LR schedule, density-grid refresh): a small
wavelet-triplane field is fitted to an analytic scene (a shaded opaque sphere on a black background) and the
loss must fall by an order of magnitude within a few hundred steps."""
import numpy as np
import pytest
import torch

from trinerflet_amd import synthetic

pytestmark = pytest.mark.gpu


def _scene_colors(o, d, radius=0.6):
    """Analytic ground truth: Lambert-ish shaded sphere, background 0."""
    b = (o * d).sum(-1)
    c = (o * o).sum(-1) - radius ** 2
    disc = b * b - c
    hit = disc > 0
    t = -b - np.sqrt(np.clip(disc, 0, None))
    p = o + t[:, None] * d
    n = p / radius
    col = 0.5 + 0.5 * n
    return (col * hit[:, None]).astype(np.float32)


def test_fits_a_sphere(cuda):
    from trinerflet_amd.nerf.network import NeRFNetwork
    from trinerflet_amd.train import TrainStep
    torch.manual_seed(0)
    m = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, density_thresh=10, hidden_dim=64,
                    hidden_dim_color=64, triplane_channels=16, triplane_resolution=128, triplane_wavelet_levels=2,
                    wavelet_type="bior6.8").to(cuda)
    iters = 400
    ts = TrainStep(m, lr=1e-2, wavelet_regularization=0.05, iters=iters, warmup_steps=0, fp16=True)
    poses = synthetic.hemisphere_poses(40, seed=1)
    rng = np.random.default_rng(0)
    N = 4096
    losses = []
    for it in range(iters):
        flat = rng.integers(0, 40 * 800 * 800, size=N)   # (choice(replace=False) permutes 25.6 M entries per call)
        pix = np.stack([flat // (800 * 800), flat % (800 * 800)], -1)
        o, d = synthetic.get_rays(poses, pix)
        gt = _scene_colors(o, d)
        t = lambda a: torch.from_numpy(a).to(cuda)
        loss = ts.step(t(o), t(d), t(gt))
        if it % 20 == 0 or it == iters - 1:
            losses.append(float(ts.last["mse"]))
    first, last = losses[0], np.mean(losses[-3:])
    assert np.isfinite(losses).all()
    assert last < 0.1 * first, (first, last, losses)
    # the occupancy grid has pruned most of the volume and the sample budget adapted
    occ = float((m.density_grid > min(m.mean_density, m.density_thresh)).float().mean())
    assert 0.0 < occ < 0.5 and m.mean_count > 0
    # the rendered image of held-out rays is close to the analytic scene: PSNR > 20 dB
    flat = rng.integers(0, 40 * 800 * 800, size=8192)
    pix = np.stack([flat // (800 * 800), flat % (800 * 800)], -1)
    o, d = synthetic.get_rays(poses, pix)
    m.eval()
    with torch.no_grad():
        out = m.render(torch.from_numpy(o).to(cuda)[None], torch.from_numpy(d).to(cuda)[None], staged=True, bg_color=0,
                       perturb=False)
    mse = float(((out["image"][0].cpu().numpy() - _scene_colors(o, d)) ** 2).mean())
    assert -10 * np.log10(mse) > 20.0, mse
