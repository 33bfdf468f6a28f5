"""CPU-PyTorch baseline: the reference's pure-PyTorch operator set timed on the host cores.

TEST INFRASTRUCTURE / REPORTED BASELINE ONLY (bench.py's `cpu_baseline` leg; never the product path).
Semantics = BASELINE.md section 3: NeRFRenderer.run (reconstruction/nerf/renderer.py:126-254: 512 uniform
steps per ray, torch cumprod compositing, colour MLP where weight > 1e-4) with this repo's near/far; planes
from depthwise conv_transpose2d (oracle/field.py::idwt_level_torch); F.grid_sample; bias-free Linear;
MSE + wavelet-L1; torch.optim.Adam(lr 1e-2, betas (0.9,0.99), eps 1e-15).  fp32, all host threads.
"""
import time

import numpy as np
import torch

from . import cref, field as ofield


def _params(C, R, scale, H, seed=0):
    g = torch.Generator().manual_seed(seed)
    J = int(round(np.log2(scale)))
    base = R // scale
    ll = (0.1 * torch.randn(3, C, base, base, generator=g)).requires_grad_(True)
    coefs = [(0.02 * 2.0 ** (-i) * torch.randn(3, C, 3, base * 2 ** i, base * 2 ** i, generator=g)).requires_grad_(True)
             for i in range(J)]
    shapes = [(H, 3 * C), (16, H), (H, 31), (H, H), (3, H)]
    W = [((torch.rand(s, generator=g) * 2 - 1) / np.sqrt(s[1])).requires_grad_(True) for s in shapes]
    return ll, coefs, W


def render_run(planes, W, rays_o, rays_d, nears, fars, bound, num_steps=512, bg=0.0, full=False):
    """NeRFRenderer.run restated (renderer.py:126-254, upsample_steps=0, perturb off).  Pinned against the reference's
    own run() by tests/golden/network_reference.npz (tests/test_reference_pins.py).  full=True also returns the
    depth (renderer.py:227-228) and weights_sum."""
    N = rays_o.shape[0]
    nears, fars = nears.unsqueeze(-1), fars.unsqueeze(-1)
    z = torch.linspace(0.0, 1.0, num_steps).unsqueeze(0).expand(N, num_steps)
    z = nears + (fars - nears) * z
    sample_dist = (fars - nears) / num_steps
    xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z.unsqueeze(-1)
    xyzs = xyzs.clamp(-bound, bound)
    feats = ofield.triplane_features(planes, xyzs.reshape(-1, 3), bound)
    h = torch.relu(torch.nn.functional.linear(feats, W[0]))
    o = torch.nn.functional.linear(h, W[1])
    sigma = ofield._TruncExp.apply(o[:, 0]).view(N, num_steps)
    deltas = torch.cat([z[..., 1:] - z[..., :-1], sample_dist * torch.ones_like(z[..., :1])], -1)
    alphas = 1 - torch.exp(-deltas * sigma)
    shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], -1)
    weights = alphas * torch.cumprod(shifted, -1)[..., :-1]
    mask = (weights > 1e-4).reshape(-1)
    dirs = rays_d.view(-1, 1, 3).expand_as(xyzs).reshape(-1, 3)
    rgbs = torch.zeros(N * num_steps, 3)
    if mask.any():
        zc = torch.cat([ofield.sh4(dirs[mask]), o[mask][:, 1:]], -1)
        hc = torch.relu(torch.nn.functional.linear(zc, W[2]))
        hc = torch.relu(torch.nn.functional.linear(hc, W[3]))
        rgbs = rgbs.index_put((mask.nonzero().squeeze(-1),), torch.sigmoid(torch.nn.functional.linear(hc, W[4])))
    rgbs = rgbs.view(N, num_steps, 3)
    ws = weights.sum(-1)
    image = (weights.unsqueeze(-1) * rgbs).sum(-2) + (1 - ws).unsqueeze(-1) * bg
    if full:
        depth = (weights * ((z - nears) / (fars - nears)).clamp(0, 1)).sum(-1)
        return {"image": image, "depth": depth, "weights_sum": ws}
    return image


def time_step(C, R, scale, H, N, wave="bior6.8", lam=0.4, bound=1.5, threads=None, repeats=3, warmup=1):
    """Full optimisation steps (rebuild planes + render + backward + Adam) at the given size: `warmup` untimed steps,
    then the MEDIAN over `repeats` timed ones (BASELINE.md section 3's procedure) of the seconds spent in the dense part
    (planes forward + backward, regulariser, Adam -- independent of the ray count) and in the per-ray part (render,
    loss, backward down to the plane gradient -- proportional to N up to the plane gradient's zero fill)."""
    from trinerflet_amd import synthetic
    if threads:
        torch.set_num_threads(threads)
    ll, coefs, W = _params(C, R, scale, H)
    opt = torch.optim.Adam([ll] + coefs + W, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    o, d = synthetic.training_rays(N, n_cams=8, seed=0)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = cref.near_far_from_aabb(o, d, aabb, 0.2)
    gt = torch.from_numpy(synthetic.target_colors(d))
    o, d, nears, fars = map(torch.from_numpy, (o, d, nears, fars))
    res = {"dense_s": [], "ray_s": []}
    for it in range(warmup + repeats):
        opt.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        planes = ofield.build_planes_torch(ll, coefs, wave)
        t1 = time.perf_counter()
        pl = planes.detach().requires_grad_(True)
        image = render_run(pl, W, o, d, nears, fars, bound)
        mse = ((image - gt) ** 2).mean()
        mse.backward()
        t2 = time.perf_counter()
        reg = ofield.wavelet_reg(coefs, lam)
        planes.backward(pl.grad, retain_graph=True)
        reg.backward()
        opt.step()
        t3 = time.perf_counter()
        del planes, pl, image, mse, reg
        if it >= warmup:
            res["dense_s"].append((t1 - t0) + (t3 - t2))
            res["ray_s"].append(t2 - t1)
    out = {k: float(np.median(v)) for k, v in res.items()}
    out["dense_all_s"], out["ray_all_s"] = res["dense_s"], res["ray_s"]
    return out
