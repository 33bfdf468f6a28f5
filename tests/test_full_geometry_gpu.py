"""BASELINE.json's configurations at their REAL geometry (README.md:46-58): base / base-light (C = 32, R = 2048,
scale 32 = five levels of bior6.8, hidden 64, 60 000 rays) and large (C = 48, hidden 128) -- the size-dependent code
(pipelined IDWT kernels, occupancy window, gradient-support rectangles, 49 k tile bins, > 4 GB buffers, the two-launch
hidden-128 backward inside TrainStep) against the oracle on what the oracle can afford:

  * the march of all 60 000 rays: per-ray (id, offset, count), counter, every sample position / step bit-exact;
  * five-level IDWT + adjoint on three (plane, channel) slices of 2048^2 vs the C oracle;
  * fused field forward / backward on a 100 000-sample subset of the marched samples vs oracle/field.py;
  * TrainStep with the occupancy window + support chain == TrainStep on whole planes, from an untrained budget
    (mean_count = 0 first) through a grid refresh.
Skipped when the device has less than 64 GB free."""
import copy

import numpy as np
import pytest
import torch

from oracle import cref, field as ofield
from trinerflet_amd import synthetic

pytestmark = pytest.mark.gpu

CONFIGS = {"base": (32, 64, 0.4), "large": (48, 128, 0.6), "small": (16, 64, 0.2), "small1": (16, 64, 0.2)}
# README.md:46-58: plane resolution, --triplane_wavelet_levels (the ratio to the 64^2 LL plane), rays per step of the stage
# the configuration ENDS in; "small1" is small's first stage (512 res, scale 8, 20 000 rays = BASELINE config 1's geometry)
GEOM = {"base": (2048, 32, 60000), "large": (2048, 32, 60000), "small": (1024, 16, 60000), "small1": (512, 8, 20000)}
R, SCALE, N, BOUND = 2048, 32, 60000, 1.5
ALL = ["base", "large", "small", "small1"]


def _need_memory():
    free, _ = torch.cuda.mem_get_info()
    if free < 64 * 2 ** 30:
        pytest.skip(f"needs 64 GB of free device memory, {free / 2 ** 30:.0f} GB available")


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _model(dev, cfg, seed=0, **kw):
    from trinerflet_amd.nerf.network import NeRFNetwork
    C, H, _ = CONFIGS[cfg]
    Rc, sc, _ = GEOM[cfg]
    m = NeRFNetwork(encoding="triplane_wavelet", bound=BOUND, cuda_ray=True, density_thresh=10, hidden_dim=H,
                    hidden_dim_color=H, triplane_channels=C, triplane_resolution=Rc, triplane_wavelet_levels=sc,
                    wavelet_type="bior6.8", **kw).to(dev)
    synthetic.init_field_parameters(m, seed=seed)
    assert [p.shape[-1] for p in m.encoder.planes_features_wavelet_coefs] == [64 << i for i in range(int(np.log2(sc)))]
    return m


def _rays(rays60k, cfg):
    """The configuration's batch: the first N rays of the 60 000 (a seeded permutation of the pixel pool)."""
    o, d, noise, bf = rays60k
    n = GEOM[cfg][2]
    return np.ascontiguousarray(o[:n]), np.ascontiguousarray(d[:n]), np.ascontiguousarray(noise[:n]), bf


@pytest.fixture(scope="module")
def rays60k():
    rng = np.random.default_rng(0)
    poses = synthetic.hemisphere_poses(100, seed=0)
    flat = rng.permutation(100 * 800 * 800)[:N]
    pix = np.stack([flat // (800 * 800), flat % (800 * 800)], -1)
    o, d = synthetic.get_rays(poses, pix)
    noise = rng.random(N).astype(np.float32)
    bf = synthetic.sphere_bitfield(128, 2, BOUND, 0.8, 0.0)
    return o, d, noise, bf


@pytest.mark.parametrize("cfg", ["base", "small1"])
def test_march_60k_rays_bit_exact(cuda, rays60k, cfg):
    """All rays of a step (60 000; small's first stage: 20 000, README.md:46) against the C oracle, bit for bit."""
    from trinerflet_amd import raymarching
    o, d, noise, bf = _rays(rays60k, cfg)
    N = GEOM[cfg][2]
    aabb = np.array([-BOUND] * 3 + [BOUND] * 3, np.float32)
    nears, fars = cref.near_far_from_aabb(o, d, aabb, 0.2)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    # first pass: unknown budget (the wrapper's .item() path); second: the budget the running mean would give
    counter = torch.zeros(2, dtype=torch.int32, device=cuda)
    x, dd, dl, rr = raymarching.march_rays_train(t(o), t(d), BOUND, t(bf), 2, 128, t(nears), t(fars), counter, -1, True,
                                                 128, False, 0, 1024, t(noise))
    total = int(counter[0])
    assert 3_000_000 * N // 60000 < total < 8_000_000 * N // 60000         # the bench's workload (~4.65 M samples at 60 000)
    xr, dr, lr, rro, cr = cref.march_rays_train(o, d, BOUND, bf, 2, 128, nears, fars, noise, total + 128)
    assert np.array_equal(counter.cpu().numpy(), cr) and np.array_equal(rr.cpu().numpy(), rro)
    assert np.array_equal(x[:total].cpu().numpy(), xr[:total]) and np.array_equal(dl[:total].cpu().numpy(), lr[:total])
    assert np.array_equal(dd[:total].cpu().numpy(), dr[:total])
    budget = total * 9 // 10                                   # a budget that drops the last rays
    counter.zero_()
    x2, _, dl2, rr2 = raymarching.march_rays_train(t(o), t(d), BOUND, t(bf), 2, 128, t(nears), t(fars), counter, budget,
                                                   True, 128, False, 0, 1024, t(noise))
    M = x2.shape[0]
    xr, dr, lr, rro, cr = cref.march_rays_train(o, d, BOUND, bf, 2, 128, nears, fars, noise, M)
    assert np.array_equal(rr2.cpu().numpy(), rro) and np.array_equal(counter.cpu().numpy(), cr)
    assert np.array_equal(x2.cpu().numpy(), xr) and np.array_equal(dl2.cpu().numpy(), lr)


@pytest.mark.parametrize("cfg", ALL)
def test_five_level_idwt_and_adjoint_slices(cuda, cfg):
    """(five levels at base / large, four at small's final stage, three at its first)"""
    _need_memory()
    C = CONFIGS[cfg][0]
    R, J = GEOM[cfg][0], int(np.log2(GEOM[cfg][1]))
    m = _model(cuda, cfg, seed=1)
    enc = m.encoder
    with torch.no_grad():
        for p in enc.planes_features_wavelet_coefs:
            p.mul_(4.0)
    planes = enc.get_planes()
    assert tuple(planes.shape) == (3, C, R, R)
    slices = [(0, 0), (1, C // 2), (2, C - 1)]
    ll = np.stack([enc.planes_features[p, c].detach().cpu().numpy() for p, c in slices])[None]
    coefs = [np.stack([q[p, c].detach().cpu().numpy() for p, c in slices])[None]
             for q in enc.planes_features_wavelet_coefs]
    want = cref.build_planes(ll, coefs, "bior6.8")[0]
    for k, (p, c) in enumerate(slices):
        got = planes[p, c].detach().cpu().numpy()
        assert np.abs(got - want[k]).max() < 5e-6 * np.abs(want[k]).max(), (cfg, p, c)
    # adjoint through autograd of the module path (the kernels TrainStep also runs)
    g = torch.Generator(device=cuda).manual_seed(3)
    cot = torch.zeros_like(planes)
    for p, c in slices:
        cot[p, c] = torch.randn(R, R, generator=g, device=cuda)
    planes.backward(cot)
    dpl = np.stack([cot[p, c].cpu().numpy() for p, c in slices])[None]
    dll, dco = cref.build_planes_adj(dpl, J, "bior6.8")
    got_ll = np.stack([enc.planes_features.grad[p, c].cpu().numpy() for p, c in slices])
    assert _rel(got_ll, dll[0]) < 2e-5
    for lvl, q in enumerate(enc.planes_features_wavelet_coefs):
        got = np.stack([q.grad[p, c].cpu().numpy() for p, c in slices])
        assert _rel(got, dco[lvl][0]) < 2e-5, (cfg, lvl)
    # slices that received no cotangent get exactly zero
    assert float(enc.planes_features_wavelet_coefs[J - 1].grad[0, 1].abs().max()) == 0.0


@pytest.mark.parametrize("cfg", ALL)
def test_fused_field_on_marched_subset(cuda, rays60k, cfg):
    _need_memory()
    from trinerflet_amd import raymarching
    from trinerflet_amd.nerf import field as gfield
    C, H, _ = CONFIGS[cfg]
    R = GEOM[cfg][0]
    o, d, noise, bf = _rays(rays60k, cfg)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    m = _model(cuda, cfg, seed=2)
    with torch.no_grad():
        for p in m.encoder.planes_features_wavelet_coefs:
            p.mul_(4.0)
    m.density_bitfield.copy_(t(bf))
    nears, fars = raymarching.near_far_from_aabb(t(o), t(d), m.aabb_train, 0.2)
    counter = torch.zeros(2, dtype=torch.int32, device=cuda)
    x, dd, dl, rr = raymarching.march_rays_train(t(o), t(d), BOUND, t(bf), 2, 128, nears, fars, counter, -1, True, 128,
                                                 False, 0, 1024, t(noise))
    total = int(counter[0])
    pick = torch.from_numpy(np.sort(np.random.default_rng(1).choice(total, 100_000, replace=False))).to(cuda)
    xs, ds = x[pick].contiguous(), dd[pick].contiguous()
    planes = m.encoder.get_planes().detach()
    tm = m.encoder.get_planes_texel_major()                          # fp16 [3,R,R,C]
    Ws = [m.sigma_net[0].weight, m.sigma_net[1].weight, m.color_net[0].weight, m.color_net[1].weight,
          m.color_net[2].weight]
    packed = gfield.pack_weights(*Ws, C, H)
    Mx = xs.shape[0]
    sigma, rgb, feats = gfield.field_forward(tm, xs, ds, packed, BOUND, C, R, H, save_feats=True)
    # oracle, fp16 operand emulation (fp32 accumulate), planes rounded to fp16 like the sampler's copy
    pl_o = planes.cpu().requires_grad_(True)
    W_o = [w.detach().cpu().clone().requires_grad_(True) for w in Ws]
    s_o, c_o = ofield.field(pl_o, xs.cpu(), ds.cpu(), W_o, BOUND, fp16=True, plane_half=True)
    np.testing.assert_allclose(rgb.cpu().numpy(), c_o.detach().numpy(), rtol=0, atol=1e-3)
    np.testing.assert_allclose(sigma.cpu().numpy(), s_o.detach().numpy(), rtol=3e-3, atol=1e-6)
    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(Mx, generator=g), torch.randn(Mx, 3, generator=g)
    ((s_o * a).sum() + (c_o * b).sum()).backward()
    # kernel backward, TrainStep's form: dF as fp16 + tile-sorted reduction into (3,C,R,R)
    nW = sum(w.numel() for w in Ws)
    gW = torch.zeros(nW, device=cuda)
    dfeat = torch.empty(3, Mx, C, dtype=torch.float16, device=cuda)
    g_cm = torch.empty(3, C, R, R, device=cuda)
    gfield.field_backward(a.to(cuda), b.to(cuda), sigma, None, feats, xs, ds, packed, BOUND, C, R, H, g_cm, gW,
                          dfeat=dfeat)
    gfield.plane_grad_binned(dfeat, xs, BOUND, C, R, g_cm, channel_major=True)
    off = 0
    for k, w in enumerate(W_o):
        e = _rel(gW[off:off + w.numel()].cpu().numpy().reshape(w.shape), w.grad.numpy())
        assert e < 5e-3, (cfg, k, e)
        off += w.numel()
    ref = pl_o.grad.numpy()
    got = g_cm.cpu().numpy()
    assert _rel(got, ref) < 5e-3
    assert np.array_equal(got != 0, ref != 0) or float(np.abs(got[ref == 0]).max()) == 0.0


@pytest.mark.parametrize("cfg", ALL)
def test_trainstep_window_equals_whole_plane(cuda, rays60k, cfg):
    """Five steps from an untrained sample budget (mean_count = 0: the march's worst-case buffers and the .item()
    path), one grid refresh at step 4; with the occupancy window + support rectangles vs whole planes.
    small (R = 1024: side_caps = (0, 0), one column-walk level) and its first stage (R = 512, 20 000 rays: no column-walk
    level, 12.6 M coefficients = below defer_adam's default threshold) take the branches base / large do not."""
    _need_memory()
    from trinerflet_amd.train import TrainStep
    C, H, lam = CONFIGS[cfg]
    R, _, N = GEOM[cfg]
    o, d, noise, bf = _rays(rays60k, cfg)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    gt = t(synthetic.target_colors(d))
    o_t, d_t, nz, bf_t = t(o), t(d), t(noise), t(bf)
    base = _model(cuda, cfg, seed=4)
    base.density_bitfield.copy_(bf_t)
    res = []
    for use_roi in (False, False, True):      # the whole-plane run twice: the yardstick for run-to-run noise
        m = copy.deepcopy(base)
        ts = TrainStep(m, lr=1e-2, wavelet_regularization=lam, iters=1000, update_extra_interval=4, use_roi=use_roi)
        ts.post_refresh = lambda m=m: m.density_bitfield.copy_(bf_t)
        m.mean_count = 0
        losses, mses, Ms = [], [], []
        for it in range(5):
            loss = ts.step(o_t, d_t, gt, noises=nz)
            losses.append(float(loss))
            mses.append(float(ts.last["mse"]))
            Ms.append(int(ts.last["counter"][0]))
            if use_roi and it % 4 != 0:
                assert ts._roi is not None and ts._roi[6] < R and ts._rect_ok and ts._rects[0] is not None
                # the r = 0.8 sphere's window: 0.53 of the plane's side + the cells' and the footprint's margin, in 64s
                assert ts._roi[6] == ts._roi[7] == {2048: 1152, 1024: 640, 512: 384}[R], ts._roi
        assert all(np.isfinite(losses)) and Ms[0] > 3_000_000 * N // 60000 and len(set(Ms)) == 1
        assert float(ts.last["found_inf"]) == 0.0
        # at this size the windowed run defers the optimiser pass outside the live rectangles (defer_adam): a step's
        # loss then carries the L1 value of the live coefficients only, the rest arrives with the replay
        assert ts.defer_adam == (bool(use_roi) and 3 * C * R * R >= 32_000_000)
        assert ts.side_caps == ((0, 0) if 3 * C * R * R < (1 << 27) else ts.side_caps) and (ts.side_caps != (0, 0)) == (R == 2048)
        total = sum(losses) + float(ts.pop_deferred_reg())
        res.append((losses, [p.detach() for p in m.parameters()], mses, total))
        del ts
    np.testing.assert_allclose(res[0][2], res[2][2], rtol=2e-4)          # MSE per step
    np.testing.assert_allclose(res[0][3], res[2][3], rtol=2e-4)          # sum of the losses incl. the deferred L1 share
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=2e-4)
    assert res[0][0][-1] < res[0][0][0]
    # The tile reduction sums in the order its bin-fill atomics produced, so two runs of the SAME configuration differ
    # in the last bits of a gradient, and Adam (eps = 1e-15) turns a gradient at rounding level into a +-lr step: the
    # windowed run may differ from the whole-plane run by no more than two whole-plane runs differ from each other
    # (x3 + a floor), never by more than a few steps of lr anywhere.
    for a, a2, b in zip(res[0][1], res[1][1], res[2][1]):
        noise = int(((a - a2).abs() > 2e-3 + 1e-3 * a2.abs()).sum())
        bad = int(((a - b).abs() > 2e-3 + 1e-3 * b.abs()).sum())
        assert bad <= 3 * noise + max(2, int(1e-5 * a.numel())) and float((a - b).abs().max()) < 6e-2, \
            (cfg, bad, noise, a.numel())


@pytest.mark.parametrize("fp32", [True, False])
def test_run_at_small_first_stage_vs_cpu_torch_baseline(cuda, rays60k, fp32):
    """BASELINE.json config 1: 'small (512 res, 16 ch, scale 8, 20k rays) pure-PyTorch path, no cuda_ray' =
    NeRFRenderer.run (renderer.py:126-254: 512 uniform steps per ray, cumprod compositing, colour where weight > 1e-4).
    renderer.run() here (HIP near/far + lookup + MLP under torch glue) at that stage's geometry against
    oracle/torch_baseline.render_run -- the restatement the CPU baseline times, pinned to the reference's own run() by
    tests/golden/network_reference.npz -- on planes the C oracle builds from the same coefficients.  2 048 of the
    stage's 20 000 rays (1 M samples on the host).  fp32: the reference's precision (fp32 planes, modular kernels);
    otherwise the default fused path (fp16 planes / MFMA operands) at the tolerance the fp32 reference fixture holds."""
    from oracle import torch_baseline
    o, d, _, _ = _rays(rays60k, "small1")
    n = 2048
    o, d = o[:n], d[:n]
    m = _model(cuda, "small1", seed=7, **(dict(plane_dtype=torch.float32) if fp32 else {}))
    m.force_modular = fp32
    m.eval()
    with torch.no_grad():
        for p in m.encoder.planes_features_wavelet_coefs:
            p.mul_(4.0)
        # run() samples the whole box (no occupancy grid): a wider, lower logit distribution so that the rays' opacities
        # spread over 0.2 .. 0.8 instead of all ending at 0.95
        m.sigma_net[0].weight.mul_(6.0)
        w = m.sigma_net[1].weight[0]
        w.mul_(6.0).sub_(0.5 * w.abs().mean())
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    with torch.no_grad():
        out = m.run(t(o)[None], t(d)[None], num_steps=512, upsample_steps=0, bg_color=1.0, perturb=False)
    ll = m.encoder.planes_features.detach().cpu().numpy()
    coefs = [p.detach().cpu().numpy() for p in m.encoder.planes_features_wavelet_coefs]
    planes = torch.from_numpy(cref.build_planes(ll, coefs, "bior6.8"))
    W = [w.detach().cpu() for w in (m.sigma_net[0].weight, m.sigma_net[1].weight, m.color_net[0].weight,
                                    m.color_net[1].weight, m.color_net[2].weight)]
    aabb = np.array([-BOUND] * 3 + [BOUND] * 3, np.float32)
    nears, fars = cref.near_far_from_aabb(o, d, aabb, 0.2)
    with torch.no_grad():
        want = torch_baseline.render_run(planes, W, torch.from_numpy(o), torch.from_numpy(d), torch.from_numpy(nears),
                                         torch.from_numpy(fars), BOUND, 512, bg=1.0, full=True)
    image, depth, ws = want["image"], want["depth"], want["weights_sum"]
    got_i, got_w = out["image"][0].cpu().numpy(), out["weights_sum"].reshape(-1).cpu().numpy()
    ws = ws.numpy()
    assert 0.05 * n < (ws > 0.5).sum() < 0.95 * n and ws.max() - ws.min() > 0.5, np.percentile(ws, [0, 5, 50, 95, 100])
    err_i, err_w = np.abs(got_i - image.numpy()).max(), np.abs(got_w - ws).max()
    print(f"run() at 512^2 x 16ch, 512 steps, fp32={fp32}: max|d image| {err_i:.2e}, max|d weights_sum| {err_w:.2e}")
    # fp32: rounding of 512-term sums for weights_sum and depth (3e-6 measured); the image also carries the colour mask's
    # threshold (renderer.py:216: colour only where weight > 1e-4 -- a sample whose weight sits at the threshold is in the
    # mask on one side and not on the other: up to 1e-4 per such sample; 5e-5 measured).  fused: 2e-3 against an fp32
    # computation (the fixture test's bound)
    tol, tol_i = (2e-5, 2e-4) if fp32 else (2e-3, 2e-3)
    assert err_i < tol_i and err_w < tol, (err_i, err_w)
    hit = np.isfinite(depth.numpy())
    assert np.abs(out["depth"][0].cpu().numpy()[hit] - depth.numpy()[hit]).max() < tol


def _render_model(cuda, bf, cfg="large"):
    """A README configuration with a medium dense enough that rays end by transmittance as well as by leaving the box."""
    m = _model(cuda, cfg, seed=6)
    m.density_bitfield.copy_(torch.from_numpy(bf).to(cuda))
    m.eval()
    # density_scale (renderer.py:309,352: sigmas = self.density_scale * sigmas) chosen so that the median optical depth
    # through the ball is about 12: most rays that cross it end by transmittance, grazing ones by leaving the box
    g = torch.Generator().manual_seed(3)
    p = torch.randn(20000, 3, generator=g)
    p = (p / p.norm(dim=-1, keepdim=True) * 0.8 * torch.rand(20000, 1, generator=g) ** (1 / 3)).to(cuda)
    with torch.no_grad():
        med = float(m.density(p)["sigma"].float().median())
    m.density_scale = float(np.float32(12.0 / med))
    return m


def test_test_render_4096_steps_large_geometry_vs_oracle_loop(cuda, rays60k):
    """BASELINE config 5's `--test` render (main_nerf.py:190-194: max_steps = 4096; renderer.py:324-374) at the LARGE
    geometry (C = 48, R = 2048, hidden 128) on a 4 096-ray subset:
      (i)  the device-driven loop against the ORACLE loop (C march_rays / composite_rays, the reference's n_step rule and
           compaction) evaluating the same field through the same HIP kernel: survivors per iteration exact, image /
           weights / depth to fp32 rounding -- march, composite, compaction and loop policy at max_steps = 4096;
      (ii) against the oracle loop with the CPU field in the kernel's operand precision (fp16 planes / operands, fp32
           accumulation): image within north_star's 1e-3, survivor counts within 0.5 % of the rays per iteration (a ray
           whose transmittance crosses T_thresh inside an fp16 rounding may end one iteration apart)."""
    _need_memory()
    from tests.test_reference_pins import oracle_infer_loop
    o, d, _, bf = rays60k
    n = 4096
    o, d = np.ascontiguousarray(o[:n]), np.ascontiguousarray(d[:n])
    m = _render_model(cuda, bf)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    bg = 1.0
    with torch.no_grad():
        out = m.render(t(o)[None], t(d)[None], staged=True, bg_color=bg, perturb=False, dt_gamma=0, max_steps=4096,
                       T_thresh=1e-4, device_loop=True)
        one = m.render(t(o)[None], t(d)[None], staged=True, bg_color=bg, perturb=False, dt_gamma=0, max_steps=4096,
                       T_thresh=1e-4)                     # the default: one persistent kernel (csrc/render.hip)
    aabb = np.array([-BOUND] * 3 + [BOUND] * 3, np.float32)
    nears, fars = cref.near_far_from_aabb(o, d, aabb, 0.2)

    def finish(ws, dep, img):
        return img + (1 - ws)[:, None] * bg, np.clip(dep - nears, 0, None) / (fars - nears)

    # (i) same field function (the HIP kernel) inside the oracle's loop
    def field_hip(x, dd):
        with torch.no_grad():
            s, c = m(t(x), t(dd))
        return (m.density_scale * s.float()).cpu().numpy(), c.float().cpu().numpy()
    ws, dep, img, hist = oracle_infer_loop(o, d, nears, fars, bf, BOUND, 4096, field_hip)
    image, depth = finish(ws, dep, img)
    ended_by_T = int(((ws > 1 - 2e-4)).sum())
    stats = (len(hist), hist[:4], int((ws > 0.5).sum()), ended_by_T, float(ws.max()), m.density_scale)
    print("large-geometry render: iterations, first survivor counts, rays with ws > 0.5, ended by T, max ws, density_scale:", stats)
    assert len(hist) > 100 and hist[0] < n and 0.05 * n < (ws > 0.5).sum() < 0.5 * n, stats
    assert ended_by_T > 50, stats                           # transmittance-terminated rays exist
    got = out["image"][0].cpu().numpy()
    np.testing.assert_allclose(got, image, rtol=0, atol=2e-5)
    np.testing.assert_allclose(one["image"][0].cpu().numpy(), image, rtol=0, atol=2e-5)        # ... and the one-kernel render
    np.testing.assert_allclose(one["weights_sum"].reshape(-1).cpu().numpy(), ws, rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["weights_sum"].reshape(-1).cpu().numpy(), ws, rtol=0, atol=2e-5)
    hit = np.isfinite(depth)
    np.testing.assert_allclose(out["depth"][0].cpu().numpy()[hit], depth[hit], rtol=0, atol=2e-5)
    # the host-driven loop reports the survivors per iteration: they must equal the oracle's, iteration by iteration
    alive_log = []
    import trinerflet_amd.raymarching as rm
    real = rm.compact_rays

    def logging_compact(rays_alive, n_alive):
        res = real(rays_alive, n_alive)
        alive_log.append(int(res[1].item()))
        return res
    rm.compact_rays = logging_compact
    try:
        with torch.no_grad():
            host = m.render(t(o)[None], t(d)[None], staged=True, bg_color=bg, perturb=False, dt_gamma=0, max_steps=4096,
                            T_thresh=1e-4, device_loop=False)
    finally:
        rm.compact_rays = real
    assert alive_log == hist, (len(alive_log), len(hist), [(i, a, b) for i, (a, b) in enumerate(zip(alive_log, hist)) if a != b][:5])
    assert torch.equal(host["image"], out["image"]) and torch.equal(host["weights_sum"], out["weights_sum"])

    # (ii) the oracle's own field in the kernel's operand precision
    planes = m.encoder.get_planes().detach().cpu()
    W = [w.detach().cpu() for w in (m.sigma_net[0].weight, m.sigma_net[1].weight, m.color_net[0].weight,
                                    m.color_net[1].weight, m.color_net[2].weight)]
    pl16 = planes.half().float()

    def field_cpu(x, dd):
        with torch.no_grad():
            s, c = ofield.field(pl16, torch.from_numpy(x), torch.from_numpy(dd), W, BOUND, fp16=True)
        return (m.density_scale * s).numpy(), c.numpy()
    ws2, dep2, img2, hist2 = oracle_infer_loop(o, d, nears, fars, bf, BOUND, 4096, field_cpu)
    image2, depth2 = finish(ws2, dep2, img2)
    err = np.abs(got - image2).max()
    assert err < 1e-3, err
    assert np.abs(out["weights_sum"].reshape(-1).cpu().numpy() - ws2).max() < 1e-3
    k = min(len(hist), len(hist2))
    assert abs(len(hist) - len(hist2)) <= 2 and np.abs(np.array(hist[:k]) - np.array(hist2[:k])).max() <= 0.005 * n


def test_test_render_800x800_device_loop_equals_host_loop_large_geometry(cuda):
    """The full 800 x 800 image of BASELINE config 5 at max_steps = 4096, large geometry: the device-driven loop (state on
    the device, iterations enqueued back to back) produces the image of the host-driven loop (one survivor count read back
    per iteration, the reference's structure) bit for bit."""
    _need_memory()
    bf = synthetic.sphere_bitfield(128, 2, BOUND, 0.8, 0.0)
    m = _render_model(cuda, bf)
    poses = synthetic.hemisphere_poses(3, seed=7)
    pix = np.stack([np.full(640000, 1), np.arange(640000)], -1)
    o, d = synthetic.get_rays(poses, pix)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    with torch.no_grad():
        dev = m.render(t(o)[None], t(d)[None], staged=True, bg_color=1.0, perturb=False, max_steps=4096, T_thresh=1e-4,
                       device_loop=True)
        one = m.render(t(o)[None], t(d)[None], staged=True, bg_color=1.0, perturb=False, max_steps=4096, T_thresh=1e-4)
        host = m.render(t(o)[None], t(d)[None], staged=True, bg_color=1.0, perturb=False, max_steps=4096, T_thresh=1e-4,
                        device_loop=False)
    assert torch.equal(dev["image"], host["image"])
    assert torch.equal(dev["weights_sum"], host["weights_sum"])
    assert torch.equal(torch.nan_to_num(dev["depth"]), torch.nan_to_num(host["depth"]))
    # the one-kernel render (the default): the same image to fp32 rounding, most pixels to the bit
    assert float((one["image"] - dev["image"]).abs().max()) < 5e-6
    assert float((one["weights_sum"] == dev["weights_sum"]).float().mean()) > 0.95
    ws = dev["weights_sum"].reshape(-1)
    assert 0.1 < float((ws > 0.5).float().mean()) < 0.4                # the ball covers about a quarter of the image


@pytest.mark.parametrize("cfg", ["base", "large", "small"])
def test_one_kernel_render_at_readme_geometry_vs_oracle_loop(cuda, rays60k, cfg):
    """The DEFAULT of run_cuda's eval branch -- render_mode="kernel", csrc/render.hip k_render_rays, for hidden 128 its
    one-workgroup-per-CU form -- at the README geometries (base: C 32 / hidden 64, large: C 48 / hidden 128; R = 2048,
    max_steps = 4096; reference: renderer.py:324-374 over raymarching.cu:701-905):
      (i)   a 4 096-ray subset against the ORACLE loop (C march_rays / composite_rays, the reference's n_step rule)
            evaluating the same HIP field: image / weights / depth to 2e-5 -- the quantities that do not depend on the
            loop's schedule;
      (ii)  against the oracle loop with its own CPU field in the kernel's operand precision: 1e-3 (north_star);
      (iii) the full 800 x 800 image against the device-driven loop: fp32 rounding everywhere, >= 95 % of the rays to the bit."""
    _need_memory()
    from tests.test_reference_pins import oracle_infer_loop
    o, d, _, bf = rays60k
    n = 4096
    o, d = np.ascontiguousarray(o[:n]), np.ascontiguousarray(d[:n])
    m = _render_model(cuda, bf, cfg)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    bg = 1.0
    with torch.no_grad():
        one = m.render(t(o)[None], t(d)[None], staged=True, bg_color=bg, perturb=False, dt_gamma=0, max_steps=4096,
                       T_thresh=1e-4, render_mode="kernel")
    aabb = np.array([-BOUND] * 3 + [BOUND] * 3, np.float32)
    nears, fars = cref.near_far_from_aabb(o, d, aabb, 0.2)

    def finish(ws, dep, img):
        with np.errstate(invalid="ignore", divide="ignore"):
            return img + (1 - ws)[:, None] * bg, np.clip(dep - nears, 0, None) / (fars - nears)

    def field_hip(x, dd):
        with torch.no_grad():
            s, c = m(t(x), t(dd))
        return (m.density_scale * s.float()).cpu().numpy(), c.float().cpu().numpy()
    ws, dep, img, hist = oracle_infer_loop(o, d, nears, fars, bf, BOUND, 4096, field_hip)
    image, depth = finish(ws, dep, img)
    assert len(hist) > 100 and (ws > 1 - 2e-4).sum() > 50 and 0.05 * n < (ws > 0.5).sum() < 0.5 * n
    got_i, got_w = one["image"][0].cpu().numpy(), one["weights_sum"].reshape(-1).cpu().numpy()
    got_d = one["depth"][0].cpu().numpy()
    np.testing.assert_allclose(got_i, image, rtol=0, atol=2e-5)
    np.testing.assert_allclose(got_w, ws, rtol=0, atol=2e-5)
    hit = np.isfinite(depth)
    np.testing.assert_allclose(got_d[hit], depth[hit], rtol=0, atol=2e-5)
    assert np.isnan(got_d[~hit]).all()
    # (ii) the oracle's own field, fp16 planes / operands, fp32 accumulation
    planes = m.encoder.get_planes().detach().cpu()
    W = [w.detach().cpu() for w in (m.sigma_net[0].weight, m.sigma_net[1].weight, m.color_net[0].weight,
                                    m.color_net[1].weight, m.color_net[2].weight)]
    pl16 = planes.half().float()
    del planes

    def field_cpu(x, dd):
        with torch.no_grad():
            s, c = ofield.field(pl16, torch.from_numpy(x), torch.from_numpy(dd), W, BOUND, fp16=True)
        return (m.density_scale * s).numpy(), c.numpy()
    ws2, dep2, img2, _ = oracle_infer_loop(o, d, nears, fars, bf, BOUND, 4096, field_cpu)
    image2, _ = finish(ws2, dep2, img2)
    assert np.abs(got_i - image2).max() < 1e-3 and np.abs(got_w - ws2).max() < 1e-3
    # (iii) the whole image
    poses = synthetic.hemisphere_poses(3, seed=7)
    pix = np.stack([np.full(640000, 1), np.arange(640000)], -1)
    fo, fd = synthetic.get_rays(poses, pix)
    with torch.no_grad():
        dev = m.render(t(fo)[None], t(fd)[None], staged=True, bg_color=bg, perturb=False, max_steps=4096, T_thresh=1e-4,
                       render_mode="device_loop")
        ker = m.render(t(fo)[None], t(fd)[None], staged=True, bg_color=bg, perturb=False, max_steps=4096, T_thresh=1e-4)
    assert float((ker["image"] - dev["image"]).abs().max()) < 5e-6
    assert float((ker["weights_sum"] - dev["weights_sum"]).abs().max()) < 5e-6
    assert float((ker["weights_sum"] == dev["weights_sum"]).float().mean()) > 0.95
    # (all three colour channels to the bit: fewer -- the loop adds a ray's samples to the image in groups of n_step across
    #  iterations, the kernel in one running sum per ray; 82 % measured at base)
    assert float((ker["image"] == dev["image"]).all(-1).float().mean()) > 0.7
    dk, dd_ = torch.nan_to_num(ker["depth"], nan=-1.0), torch.nan_to_num(dev["depth"], nan=-1.0)
    assert float((dk - dd_).abs().max()) < 2e-5


@pytest.mark.parametrize("cfg", ["base", "large"])
def test_one_kernel_render_stops_at_exactly_max_steps(cuda, rays60k, cfg):
    """A ray still alive after max_steps samples: the one-kernel render composites exactly max_steps samples (the alive-ray
    loop's cap depends on its schedule: max_steps .. max_steps + 7, renderer.py:338-372 -- stated in INTEGRATION.md).  The
    oracle loop run ONE RAY AT A TIME takes one sample per iteration (n_step = max(min(1 // 1, 8), 1)), i.e. it stops at
    exactly max_steps samples too: the two must agree to fp32 rounding."""
    _need_memory()
    from tests.test_reference_pins import oracle_infer_loop
    _, _, _, bf_ball = rays60k
    m = _render_model(cuda, bf_ball, cfg)
    m.density_scale = m.density_scale * 0.002         # nearly transparent: no ray ends by transmittance
    # The step is dt_min = 2 sqrt(3) / max_steps (raymarching.cu:338-346): only a chord longer than 2 sqrt(3) = 3.46 through
    # OCCUPIED space holds more than max_steps samples -- every cell occupied, rays along the box's space diagonals (5.2).
    bf = np.full_like(bf_ball, 255)
    m.density_bitfield.fill_(255)
    rng = np.random.default_rng(5)
    u = np.array([[1, 1, 1], [1, 1, -1], [1, -1, 1], [-1, 1, 1]], np.float64)[rng.integers(0, 4, 16)] + 0.04 * rng.standard_normal((16, 3))
    u /= np.linalg.norm(u, axis=-1, keepdims=True)
    o16, d16 = (4.0311 * u).astype(np.float32), (-u).astype(np.float32)
    aabb = np.array([-BOUND] * 3 + [BOUND] * 3, np.float32)
    n16, f16 = cref.near_far_from_aabb(o16, d16, aabb, 0.2)
    assert ((f16 - n16) > 4.5).all()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    K = 96
    with torch.no_grad():
        cap = m.render(t(o16)[None], t(d16)[None], staged=True, bg_color=0.0, perturb=False, max_steps=K, T_thresh=1e-4,
                       render_mode="kernel")
        full = m.render(t(o16)[None], t(d16)[None], staged=True, bg_color=0.0, perturb=False, max_steps=4096, T_thresh=1e-4,
                        render_mode="kernel")

    def field_hip(x_, d_):
        with torch.no_grad():
            s, c = m(t(x_), t(d_))
        return (m.density_scale * s.float()).cpu().numpy(), c.float().cpu().numpy()
    ws = np.zeros(16, np.float32)
    img = np.zeros((16, 3), np.float32)
    for k in range(16):
        w1, _, i1, hist = oracle_infer_loop(o16[k:k + 1], d16[k:k + 1], n16[k:k + 1], f16[k:k + 1], bf, BOUND, K, field_hip)
        assert len(hist) == K and hist[-1] == 1          # alive to the cap, one sample per iteration
        ws[k], img[k] = w1[0], i1[0]
    np.testing.assert_allclose(cap["weights_sum"].reshape(-1).cpu().numpy(), ws, rtol=0, atol=2e-6)
    np.testing.assert_allclose(cap["image"][0].cpu().numpy(), img, rtol=0, atol=2e-6)
    assert torch.isfinite(full["weights_sum"]).all()


@pytest.mark.parametrize("cfg,plane_dtype", [("base", torch.float16), ("base", torch.float32), ("small", torch.float16),
                                             ("small1", torch.float16), ("small1", torch.float32)])
def test_trainstep_window_equals_whole_plane_bit_for_bit_when_deterministic(cuda, rays60k, cfg, plane_dtype):
    """With TrainStep(deterministic=True) every tile list of the plane-gradient reduction is ordered by sample id, so the
    summation order no longer depends on the fill pass's atomics: the windowed step (occupancy window, gradient-support
    rectangles, live rectangles + deferred optimiser pass) must then reproduce the whole-plane step BIT FOR BIT -- every
    parameter after five steps through a grid refresh, at the base geometry."""
    _need_memory()
    from trinerflet_amd.train import TrainStep
    C, H, lam = CONFIGS[cfg]
    R = GEOM[cfg][0]
    o, d, noise, bf = _rays(rays60k, cfg)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    gt = t(synthetic.target_colors(d))
    o_t, d_t, nz, bf_t = t(o), t(d), t(noise), t(bf)
    base = _model(cuda, cfg, seed=4, plane_dtype=plane_dtype)      # (fp32: the reference's training precision, F9)
    base.density_bitfield.copy_(bf_t)
    res = []
    for use_roi in (False, True):
        m = copy.deepcopy(base)
        ts = TrainStep(m, lr=1e-2, wavelet_regularization=lam, iters=1000, update_extra_interval=4, use_roi=use_roi,
                       deterministic=True)
        ts.post_refresh = lambda m=m: m.density_bitfield.copy_(bf_t)
        m.mean_count = 0
        mses = []
        for it in range(5):
            ts.step(o_t, d_t, gt, noises=nz)
            mses.append(float(ts.last["mse"]))
        ts.flush_deferred()
        assert ts.defer_adam == (use_roi and 3 * C * R * R >= 32_000_000) and (ts._roi is not None) == use_roi
        res.append((mses, [p.detach().clone() for p in m.parameters()]))
        del ts
    # (the reported MSE is a float-atomic sum over the rays: equal to rounding, not to the bit; the gradients do not read it)
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=2e-6)
    bad = [(tuple(a.shape), int((a != b).sum()), float((a - b).abs().max())) for a, b in zip(res[0][1], res[1][1])
           if not torch.equal(a, b)]
    assert not bad, bad
