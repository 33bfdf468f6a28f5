"""A13: one optimisation step, three ways, at toy size:
  (1) TrainStep            -- the fused step: raw C-ABI calls, flat buffers, fused Adam+L1
  (2) drop-in modules      -- NeRFNetwork.render + torch autograd + torch.optim.Adam (how main_nerf.py drives it)
  (3) CPU oracle           -- torch fp32 + the C oracle for marching/compositing
"""
import copy

import numpy as np
import pytest
import torch

from oracle import cref, field as ofield
from trinerflet_amd import synthetic

pytestmark = pytest.mark.gpu

C, R, SCALE, H, N, BOUND, LAM = 16, 64, 4, 64, 384, 1.5, 0.2


def _relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _model(dev):
    from trinerflet_amd.nerf.network import NeRFNetwork
    m = NeRFNetwork(encoding="triplane_wavelet", bound=BOUND, cuda_ray=True, density_thresh=10, hidden_dim=H,
                    hidden_dim_color=H, triplane_channels=C, triplane_resolution=R, triplane_wavelet_levels=SCALE,
                    wavelet_type="bior6.8").to(dev)
    synthetic.init_field_parameters(m, seed=3)
    with torch.no_grad():  # larger detail coefficients so the regulariser and the data term both matter
        for p in m.encoder.planes_features_wavelet_coefs:
            p.mul_(5.0)
    return m


class _OracleComposite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sig, rgb, deltas, rays):
        ws, dep, img = cref.composite_rays_train_forward(sig.numpy(), rgb.numpy(), deltas, rays, 1e-4)
        ctx.save = (sig.numpy().copy(), rgb.numpy().copy(), deltas, rays, ws, img)
        return torch.from_numpy(ws), torch.from_numpy(img)

    @staticmethod
    def backward(ctx, gws, gimg):
        sig, rgb, deltas, rays, ws, img = ctx.save
        gs, gc = cref.composite_rays_train_backward(gws.numpy(), gimg.numpy(), sig, rgb, deltas, rays, ws, img, 1e-4)
        return torch.from_numpy(gs), torch.from_numpy(gc), None, None


def test_one_step_three_ways(cuda):
    from trinerflet_amd.train import TrainStep
    o, d = synthetic.training_rays(N, n_cams=4, seed=7)
    gt = synthetic.target_colors(d)
    noise = np.random.default_rng(0).random(N).astype(np.float32)
    bf = synthetic.sphere_bitfield(128, 2, BOUND, 0.8, 0.55)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)

    base = _model(cuda)
    base.density_bitfield.copy_(t(bf))
    names = [n for n, _ in base.named_parameters()]
    init = {n: p.detach().clone() for n, p in base.named_parameters()}

    # ---------------- (3) CPU oracle
    ll = init["encoder.planes_features"].cpu().clone().requires_grad_(True)
    coefs = [init[f"encoder.planes_features_wavelet_coefs.{i}"].cpu().clone().requires_grad_(True) for i in range(2)]
    W = [init[k].cpu().clone().requires_grad_(True) for k in
         ("sigma_net.0.weight", "sigma_net.1.weight", "color_net.0.weight", "color_net.1.weight", "color_net.2.weight")]
    aabb = np.array([-BOUND] * 3 + [BOUND] * 3, np.float32)
    nears, fars = cref.near_far_from_aabb(o, d, aabb, 0.2)
    xyz, dirs, deltas, rays, counter = cref.march_rays_train(o, d, BOUND, bf, 2, 128, nears, fars, noise, N * 1024)
    total = int(counter[0])
    planes = ofield.build_planes_torch(ll, coefs, "bior6.8")
    sig, rgb = ofield.field(planes, torch.from_numpy(xyz[:total]), torch.from_numpy(dirs[:total]), W, BOUND, fp16=True,
                            plane_half=True)
    ws, img = _OracleComposite.apply(sig, rgb, deltas[:total], rays)
    mse = ((img - torch.from_numpy(gt)) ** 2).mean()           # background_color = 0
    reg = ofield.wavelet_reg(coefs, LAM)
    (mse + reg).backward()
    g_mse_ll = ll.grad.clone()
    opt = torch.optim.Adam([ll] + coefs + W, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    opt.step()

    # ---------------- (1) fused step (GradScaler default initial scale 2^16: fp16 backward operands neither underflow nor overflow)
    # ... with the optional fused form (m1f: Adam inside the adjoint IDWT kernels) and as shipped (separate optimiser
    # pass, coefficient gradients observable); both must land on the same parameters
    m1f = copy.deepcopy(base)
    tsf = TrainStep(m1f, lr=1e-2, wavelet_regularization=LAM, iters=1000, warmup_steps=0, fp16=True, init_scale=65536.0,
                    update_extra_interval=0, fuse_adam=True)
    m1f.mean_count = 0
    tsf.step(t(o), t(d), t(gt), noises=t(noise))
    m1 = copy.deepcopy(base)
    ts = TrainStep(m1, lr=1e-2, wavelet_regularization=LAM, iters=1000, warmup_steps=0, fp16=True, init_scale=65536.0,
                   update_extra_interval=0, fuse_adam=False)  # no density-grid refresh: keep the analytic bitfield
    m1.mean_count = 0
    loss1 = ts.step(t(o), t(d), t(gt), noises=t(noise))
    assert int(ts.last["counter"][0]) == total                      # bit-exact sample count vs the oracle
    assert abs(float(ts.last["mse"]) - float(mse)) < 2e-3 * float(mse)
    assert abs(float(ts.last["wavelet_reg"]) - float(reg)) < 1e-5 * float(reg)
    inv = 1.0 / 65536.0
    assert _relerr(ts.ll.grad.cpu().numpy() * inv, g_mse_ll.numpy().reshape(-1)) < 1e-2

    # ---------------- (2) drop-in modules + autograd + torch.optim.Adam
    m2 = copy.deepcopy(base)
    m2.train()
    m2.mean_count = 0
    opt2 = torch.optim.Adam(m2.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    m2.encoder.reset_cahce(); m2.encoder.get_planes()
    out = m2.render(t(o)[None], t(d)[None], staged=False, bg_color=0, perturb=True, force_all_rays=False,
                    noises=t(noise), dt_gamma=0, max_steps=1024)
    mse2 = ((out["image"][0] - t(gt)) ** 2).mean()
    wf = m2.encoder.get_wavelet_features()
    tot = sum(v.numel() for v in wf)
    reg2 = LAM * sum(v.abs().mean() * (v.numel() / tot) for v in wf) / len(wf)     # utils.py:639-655
    ((mse2 + reg2) * 65536.0).backward()          # scaler.scale(loss).backward()   (utils.py:1166)
    for p in m2.parameters():                     # scaler.step(optimizer) unscales first  (:1170)
        if p.grad is not None:
            p.grad.mul_(inv)
    opt2.step()
    assert abs(float(mse2) - float(ts.last["mse"])) < 1e-5 * float(mse2) + 1e-9
    assert _relerr(m2.encoder.planes_features.grad.cpu().numpy().reshape(-1), ts.ll.grad.cpu().numpy() * inv) < 2e-3

    # ---------------- parameters after the step: (1) == (2) and both close to (3)
    p1 = dict(m1.named_parameters())
    p2 = dict(m2.named_parameters())
    assert abs(float(tsf.last["wavelet_reg"]) - float(ts.last["wavelet_reg"])) < 1e-6 * float(ts.last["wavelet_reg"])
    for n, pf in m1f.named_parameters():   # fused optimiser == separate optimiser pass (same kernels upstream)
        g2 = p2[n].grad.detach().cpu().numpy()
        sig = np.abs(g2) > 1e-3 * np.abs(g2).max()
        assert np.mean(np.abs(pf.detach().cpu().numpy() - p1[n].detach().cpu().numpy())[sig] > 1e-6) < 1e-3, n
    oracle_new = {"encoder.planes_features": ll, "encoder.planes_features_wavelet_coefs.0": coefs[0],
                  "encoder.planes_features_wavelet_coefs.1": coefs[1], "sigma_net.0.weight": W[0],
                  "sigma_net.1.weight": W[1], "color_net.0.weight": W[2], "color_net.1.weight": W[3],
                  "color_net.2.weight": W[4]}
    # Adam's first step with eps = 1e-15 is lr * sign(g): an element can only differ between two correct
    # implementations where its gradient is at rounding-noise level, so the comparison is restricted to elements
    # whose gradient is significant (> 1e-3 of the tensor's largest); those must agree.
    for n in names:
        a, b, c0 = p1[n].detach().cpu().numpy(), p2[n].detach().cpu().numpy(), init[n].cpu().numpy()
        assert not np.array_equal(a, c0), n                              # it moved
        g2 = p2[n].grad.detach().cpu().numpy()
        sig_mask = np.abs(g2) > 1e-3 * np.abs(g2).max()
        assert sig_mask.mean() > 0.05, n
        frac12 = np.mean(np.abs(a - b)[sig_mask] > 1e-6)
        assert frac12 < 1e-3, (n, frac12)
        go = oracle_new[n].grad.detach().numpy()
        mask_o = sig_mask & (np.abs(go) > 1e-3 * np.abs(go).max())
        frac13 = np.mean(np.abs(a - oracle_new[n].detach().numpy())[mask_o] > 1e-6)
        assert frac13 < 1e-2, (n, frac13)


def test_scaler_skips_on_overflow(cuda):
    """GradScaler semantics: a non-finite gradient leaves parameters untouched and halves the scale."""
    from trinerflet_amd.train import TrainStep
    o, d = synthetic.training_rays(N, n_cams=4, seed=7)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    m = _model(cuda)
    m.density_bitfield.copy_(t(synthetic.sphere_bitfield(128, 2, BOUND, 0.8, 0.55)))
    ts = TrainStep(m, fp16=True, init_scale=2.0 ** 40, update_extra_interval=0)  # guarantees fp16 overflow
    m.mean_count = 0
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    ts.step(t(o), t(d), t(synthetic.target_colors(d)))
    assert float(ts.last["found_inf"]) == 1.0
    assert float(ts.scale) == 2.0 ** 39
    for n, p in m.named_parameters():
        assert torch.equal(p.detach(), before[n]), n


def test_train_parity_mode_fp32_planes_and_unscaled_loss(cuda):
    """`plane_dtype=torch.float32` ("train-parity mode": fp32 texels; the occupancy window applies to both precisions since round 6) and `fp16=False` (no
    GradScaler): the same step as the default fast mode up to the fp16 rounding of the texels."""
    from trinerflet_amd.nerf.network import NeRFNetwork
    from trinerflet_amd.train import TrainStep
    o, d = synthetic.training_rays(N, n_cams=4, seed=7)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    gt = t(synthetic.target_colors(d))
    noise = t(np.random.default_rng(0).random(N).astype(np.float32))
    bf = t(synthetic.sphere_bitfield(128, 2, BOUND, 0.8, 0.55))
    losses = {}
    for tag, kw, fp16 in (("fast", {}, True), ("fp32planes", dict(plane_dtype=torch.float32), True),
                          ("noscaler", {}, False)):
        torch.manual_seed(0)
        m = NeRFNetwork(encoding="triplane_wavelet", bound=BOUND, cuda_ray=True, density_thresh=10, hidden_dim=H,
                        hidden_dim_color=H, triplane_channels=C, triplane_resolution=R, triplane_wavelet_levels=SCALE,
                        wavelet_type="bior6.8", **kw).to(cuda)
        synthetic.init_field_parameters(m, seed=3)
        m.density_bitfield.copy_(bf)
        ts = TrainStep(m, lr=1e-2, wavelet_regularization=LAM, fp16=fp16, update_extra_interval=0)
        m.mean_count = 0
        ls = [float(ts.step(t(o), t(d), gt, noises=noise)) for _ in range(3)]
        assert np.isfinite(ls).all() and ls[-1] < ls[0]
        assert float(ts.last["found_inf"]) == 0.0
        if not fp16:
            assert float(ts.scale) == 1.0
        losses[tag] = ls
    np.testing.assert_allclose(losses["fp32planes"], losses["fast"], rtol=2e-3)
    np.testing.assert_allclose(losses["noscaler"], losses["fast"], rtol=2e-3)


@pytest.mark.parametrize("force_modular", [False, True])
def test_reference_loop_under_autocast(cuda, force_modular):
    """The reference wraps train_step in torch.cuda.amp.autocast(fp16) (utils.py:1162): the drop-in modules must run
    there, fused and modular (nn.Linear in fp16, everything else fp32 as the reference's custom_fwd casts dictate)."""
    m = _model(cuda)
    m.force_modular = force_modular
    m.train()
    m.density_bitfield.copy_(torch.from_numpy(synthetic.sphere_bitfield(128, 2, BOUND, 0.8, 0.55)).to(cuda))
    m.mean_count = 0
    o, d = synthetic.training_rays(N, n_cams=4, seed=7)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    gt = t(synthetic.target_colors(d))
    opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    scaler = torch.amp.GradScaler("cuda")
    losses = []
    for it in range(4):
        m.encoder.reset_cahce(); m.encoder.get_planes()
        if it == 0:
            with torch.autocast("cuda", dtype=torch.float16):
                m.update_extra_state()
            m.density_bitfield.copy_(torch.from_numpy(synthetic.sphere_bitfield(128, 2, BOUND, 0.8, 0.55)).to(cuda))
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            out = m.render(t(o)[None], t(d)[None], staged=False, bg_color=0, perturb=True, force_all_rays=True)
            loss = ((out["image"][0] - gt) ** 2).mean()
            m.encoder.reset_cahce()
            scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(float(loss))
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


@pytest.mark.gpu
def test_step_state_kernels_match_gradscaler(cuda):
    """csrc/stepstate.hip against the torch expressions it replaces: zeroing + 1/scale, the found_inf probe of
    GradScaler.unscale_, and GradScaler.update() (`torch._amp_update_scale_`) over a sequence with skipped steps."""
    import trinerflet_amd._lib as L
    lib = L.lib()
    scale = torch.tensor([1024.0], device=cuda)
    inv, abs_sum, mse = (torch.full((1,), 7.0, device=cuda) for _ in range(3))
    flag = torch.ones(1, dtype=torch.int32, device=cuda)
    g = torch.randn(13571, device=cuda)
    L.check(lib.tnl_step_prologue(L.ptr(scale), L.ptr(inv), L.ptr(abs_sum), L.ptr(flag), L.ptr(mse), L.ptr(g),
                                  L.u32(g.numel()), L.stream()), "prologue")
    assert float(inv) == 1.0 / 1024.0 and float(abs_sum) == 0 and float(mse) == 0 and int(flag) == 0
    assert not g.any()

    probe, found = torch.empty(1, device=cuda), torch.empty(1, device=cuda)

    def run_probe(g0, g1, fl):
        L.check(lib.tnl_scaler_probe(L.ptr(g0), L.u32(g0.numel()), L.ptr(g1), L.u32(0 if g1 is None else g1.numel()),
                                     L.ptr(fl), L.ptr(probe), L.ptr(found), L.stream()), "probe")
        return float(probe), float(found)
    a, b = torch.randn(13571, device=cuda), torch.randn(777, device=cuda)
    p, f = run_probe(a, b, flag)
    ref = float(a.double().abs().sum() + b.double().abs().sum())
    assert f == 0.0 and abs(p - ref) < 1e-4 * ref
    p, f = run_probe(a, None, None)
    assert f == 0.0 and abs(p - float(a.double().abs().sum())) < 1e-4 * ref
    flag.fill_(1)
    assert run_probe(a, b, flag)[1] == 1.0
    flag.zero_()
    b[5] = float("nan")
    assert run_probe(a, b, flag)[1] == 1.0
    b[5] = float("-inf")
    assert run_probe(a, b, flag)[1] == 1.0

    # GradScaler.update(): growth after `interval` clean steps, backoff and tracker reset on a skipped one
    interval = 3
    s_k, t_k = torch.tensor([65536.0], device=cuda), torch.zeros(1, dtype=torch.int32, device=cuda)
    s_t, t_t = s_k.clone(), t_k.clone()
    steps_k = torch.zeros(1, device=cuda)
    reg = torch.empty(1, device=cuda)
    abs_sum.fill_(3.5)
    seq = [0, 0, 1, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0]
    for fi in seq:
        fi_t = torch.tensor([float(fi)], device=cuda)
        L.check(lib.tnl_step_epilogue(L.ptr(fi_t), L.ptr(steps_k), L.ptr(s_k), L.ptr(t_k), L.f32(2.0), L.f32(0.5),
                                      L.i32(interval), L.i32(1), L.ptr(abs_sum), L.f32(0.25), L.ptr(reg), L.stream()),
                "epilogue")
        torch._amp_update_scale_(s_t, t_t, fi_t, 2.0, 0.5, interval)
        assert torch.equal(s_k, s_t) and torch.equal(t_k, t_t)
    assert float(steps_k) == seq.count(0) and float(reg) == 3.5 * 0.25
    L.check(lib.tnl_step_epilogue(L.ptr(fi_t), L.ptr(steps_k), L.ptr(s_k), L.ptr(t_k), L.f32(2.0), L.f32(0.5),
                                  L.i32(interval), L.i32(0), L.ptr(None), L.f32(0.25), L.ptr(reg), L.stream()), "epilogue")
    assert torch.equal(s_k, s_t) and float(reg) == 0.0     # update_scale = 0 (fp32 training): scale untouched


def test_fused_adam_l1_against_torch_optim_adam(cuda):
    """tnl_adam_l1_step (csrc/adam_common.h adam1: every operation rounded on its own, reciprocal + Newton division,
    hardware square root) against torch.optim.Adam on the same gradients + the L1 term's sign gradient, six steps,
    eps = 1e-15 as the reference configures it (main_nerf.py:119): parameters and both moments to a few ulp."""
    import trinerflet_amd._lib as L
    lib = L.lib()
    n, lr, b1, b2, eps, l1 = 1 << 20, 1e-2, 0.9, 0.99, 1e-15, 3e-7
    g = torch.Generator(device="cpu").manual_seed(8)
    p0 = (torch.randn(n, generator=g) * 0.05).to(cuda)
    p0[::13] = 0.0
    grads = [(torch.randn(n, generator=g) * 10 ** float(torch.randint(-6, 1, (1,), generator=g))).to(cuda) for _ in range(6)]
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=lr, betas=(b1, b2), eps=eps)
    p, m, v = p0.clone(), torch.zeros(n, device=cuda), torch.zeros(n, device=cuda)
    steps = torch.zeros(1, device=cuda)
    zero = torch.zeros(1, device=cuda)
    G = 0.0
    for k, gk in enumerate(grads):
        G = max(G, float(gk.abs().max()))
        ref.grad = gk + l1 * torch.sign(ref.detach())
        opt.step()
        L.check(lib.tnl_adam_l1_step_dev(L.ptr(p), L.ptr(gk.clone()), L.ptr(m), L.ptr(v), L.u64(n), L.f32(lr), L.ptr(steps),
                                         L.f32(b1), L.f32(b2), L.f32(eps), L.f32(1.0), None, L.f32(l1), L.ptr(zero), None,
                                         L.i32(0), L.stream()), "adam_l1_step_dev")
        steps += 1
        st = opt.state[ref]
        for got, want, name in ((p, ref.detach(), "p"), (m, st["exp_avg"], "m"), (v, st["exp_avg_sq"], "v")):
            err = (got - want).abs()
            # p: the update is lr * m_hat / (sqrt(v_hat) + eps) ~ lr; a few ulp of THAT (3e-6 * lr) per step on top of the parameter's own
            # m, v: torch's lerp / addcmul fuse a multiply-add here and there: an ulp of the largest term that went in
            tol = 4e-7 * want.abs() + {"p": 3e-6 * lr * (k + 1), "m": 1e-7 * G, "v": 1e-7 * G * G}[name]
            bad = int((err > tol).sum())
            assert bad == 0, (k, name, bad, float(err.max()), float((err / (want.abs() + 1e-30)).max()))


def test_deterministic_step_is_reproducible_bit_for_bit(cuda):
    """TrainStep(deterministic=True): two runs of the same six steps (a grid refresh inside, side-stream march, tile
    sort with atomics) end with identical bits in every parameter and every Adam moment; the default mode differs in
    the last bits of the plane gradient from run to run (the fill pass's atomic arrival order)."""
    import copy
    from trinerflet_amd import synthetic
    from trinerflet_amd.nerf.network import NeRFNetwork
    from trinerflet_amd.train import TrainStep
    torch.manual_seed(0)
    base = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, density_thresh=10, hidden_dim=64,
                       hidden_dim_color=64, triplane_channels=16, triplane_resolution=256, triplane_wavelet_levels=4,
                       wavelet_type="bior6.8").to(cuda)
    synthetic.init_field_parameters(base, seed=3)
    bf = torch.from_numpy(synthetic.sphere_bitfield(128, 2, 1.5, 0.8, 0.0)).to(cuda)
    base.density_bitfield.copy_(bf)
    o, d = synthetic.training_rays(8192, n_cams=8, seed=2)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    o_t, d_t, gt = t(o), t(d), t(synthetic.target_colors(d))
    nz = torch.rand(8192, device=cuda, generator=torch.Generator(device=cuda).manual_seed(1))
    runs = []
    for _ in range(2):
        m = copy.deepcopy(base)
        torch.manual_seed(7)          # the refreshes' in-cell jitter
        ts = TrainStep(m, lr=1e-2, wavelet_regularization=0.2, iters=100, update_extra_interval=4, deterministic=True)
        ts.post_refresh = lambda m=m: m.density_bitfield.copy_(bf)
        for it in range(6):
            ts.step(o_t, d_t, gt, noises=nz, next_rays=(o_t, d_t, nz))
        ts.flush_deferred()
        runs.append([p.detach().clone() for p in m.parameters()] + [ts.coef.m.clone(), ts.coef.v.clone(), ts.mlp.m.clone()])
    for a, b in zip(*runs):
        assert torch.equal(a, b), (a.shape, int((a != b).sum()))


def test_trainstep_with_wavelet_base_resolution(cuda):
    """SURVEY 8(f) rank 4 tail: TrainStep on an encoder with wavelet_base_resolution > 0 (triplane_encoder.py:391-393: the
    levels below it keep the uncropped analysis size and are synthesised without the zero halo).  The level sizes are
    then not base * 2^i: the dense rebuild crops those levels and its adjoint zero-pads.  Checked against the module path
    (the same encoder through autograd): prediction, loss and every coefficient / LL gradient of one step."""
    from trinerflet_amd.nerf.network import NeRFNetwork
    from trinerflet_amd.train import TrainStep
    torch.manual_seed(0)
    base = NeRFNetwork(encoding="triplane_wavelet", bound=BOUND, cuda_ray=True, density_thresh=10, hidden_dim=H,
                       hidden_dim_color=H, triplane_channels=C, triplane_resolution=128, triplane_wavelet_levels=8,
                       wavelet_type="bior6.8", wavelet_base_resolution=48).to(cuda)
    sizes = [p.shape[-1] for p in base.encoder.planes_features_wavelet_coefs]
    assert sizes != [16, 32, 64] and sizes[-1] == 64, sizes           # the coarse levels have the uncropped sizes
    with torch.no_grad():
        g = torch.Generator(device=cuda).manual_seed(1)
        base.encoder.planes_features.copy_(0.1 * torch.randn(base.encoder.planes_features.shape, device=cuda, generator=g))
        for i, p in enumerate(base.encoder.planes_features_wavelet_coefs):
            p.copy_(0.05 * 0.5 ** i * torch.randn(p.shape, device=cuda, generator=g))
    bf = torch.from_numpy(synthetic.sphere_bitfield(128, 2, BOUND, 0.8, 0.3)).to(cuda)
    base.density_bitfield.copy_(bf)
    o, d = synthetic.training_rays(1024, n_cams=4, seed=5)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    o_t, d_t, gt = t(o), t(d), t(synthetic.target_colors(d))
    nz = torch.rand(1024, device=cuda, generator=torch.Generator(device=cuda).manual_seed(2))
    # fused step
    m1 = copy.deepcopy(base)
    ts = TrainStep(m1, lr=1e-2, wavelet_regularization=0.0, iters=100, fp16=True, update_extra_interval=0, init_scale=65536.0)
    assert not ts.use_roi and ts.base_res == 48
    m1.mean_count = 0
    ts.step(o_t, d_t, gt, noises=nz)
    # module path
    m2 = copy.deepcopy(base)
    m2.train()
    m2.mean_count = 0
    m2.encoder.reset_cahce(); m2.encoder.get_planes()
    out = m2.render(o_t[None], d_t[None], staged=False, bg_color=0, perturb=True, force_all_rays=False, noises=nz,
                    dt_gamma=0, max_steps=1024)
    mse = ((out["image"][0] - gt) ** 2).mean()
    (mse * 65536.0).backward()          # the loss scale of GradScaler: the fp16 backward operands underflow without it
    assert np.abs(ts.last["image"].cpu().numpy() - out["image"][0].detach().cpu().numpy()).max() < 1e-5
    assert abs(float(ts.last["mse"]) - float(mse)) < 1e-5 * float(mse)
    inv = 1.0 / 65536.0
    errs = {"LL": _relerr(ts.ll.grad.cpu().numpy() * inv, m2.encoder.planes_features.grad.cpu().numpy().reshape(-1) * inv)}
    for i, p in enumerate(m2.encoder.planes_features_wavelet_coefs):
        errs[i] = _relerr(ts.coef.grad_view(i).cpu().numpy() * inv, p.grad.cpu().numpy() * inv)
        assert float(p.grad.abs().sum()) > 0
    assert all(e < 2e-2 for e in errs.values()), errs


@pytest.mark.parametrize("thr,windowed", [(64, False), (128, False), (128, True)])
def test_frozen_levels_follow_the_reference_clear_grad(cuda, thr, windowed):
    """--min_wavelet_resolution_to_learn (run_utils.py:88): Trainer.clear_grad (utils.py:1105-1114, called between backward
    and scaler.step, :1168) drops every gradient except those of the encoder parameters whose last dimension exceeds the
    threshold; torch.optim.Adam then skips the others entirely.  TrainStep(min_wavelet_resolution_to_learn=thr) against
    that loop on the drop-in modules (autograd + torch.optim.Adam), three steps: the frozen tensors (MLP weights, LL plane,
    levels with n <= thr) keep their bits in both, the learning levels agree where their gradient is significant, the
    loss (which still carries the frozen levels' L1 value) agrees.  windowed: R = 512 with an occupancy window, so that
    the live / deferred split and the adjoint's fused optimiser levels see the frozen prefix too."""
    from trinerflet_amd.nerf.network import NeRFNetwork
    from trinerflet_amd.train import TrainStep
    Rt, sc, n_rays = (512, 8, 2048) if windowed else (256, 4, 1024)
    m0 = NeRFNetwork(encoding="triplane_wavelet", bound=BOUND, cuda_ray=True, density_thresh=10, hidden_dim=H,
                     hidden_dim_color=H, triplane_channels=C, triplane_resolution=Rt, triplane_wavelet_levels=sc,
                     wavelet_type="bior6.8").to(cuda)
    synthetic.init_field_parameters(m0, seed=3)
    with torch.no_grad():
        for p in m0.encoder.planes_features_wavelet_coefs:
            p.mul_(5.0)
    o, d = synthetic.training_rays(n_rays, n_cams=4, seed=7)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    gt = t(synthetic.target_colors(d))
    noise = t(np.random.default_rng(0).random(n_rays).astype(np.float32))
    bf = t(synthetic.sphere_bitfield(128, 2, BOUND, 0.5 if windowed else 0.8, 0.0))
    m0.density_bitfield.copy_(bf)
    init = {n: p.detach().clone() for n, p in m0.named_parameters()}
    learns = lambda n, p: n.startswith("encoder.") and p.shape[-1] > thr
    # (1) the reference's loop: backward, clear_grad, Adam
    m2 = copy.deepcopy(m0)
    m2.train()
    m2.mean_count = 0
    opt2 = torch.optim.Adam(m2.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    losses2 = []
    for it in range(3):
        opt2.zero_grad(set_to_none=True)
        m2.encoder.reset_cahce(); m2.encoder.get_planes()
        out = m2.render(t(o)[None], t(d)[None], staged=False, bg_color=0, perturb=True, force_all_rays=True,
                        noises=noise, dt_gamma=0, max_steps=1024)
        mse2 = ((out["image"][0] - gt) ** 2).mean()
        wf = m2.encoder.get_wavelet_features()
        tot = sum(v.numel() for v in wf)
        reg2 = LAM * sum(v.abs().mean() * (v.numel() / tot) for v in wf) / len(wf)
        ((mse2 + reg2) * 65536.0).backward()
        wavelet_grad = [val.grad for val in m2.encoder.parameters()]          # clear_grad, utils.py:1105-1114
        for val in m2.parameters():
            val.grad = None
        for idx, val in enumerate(m2.encoder.parameters()):
            if val.shape[-1] > thr:
                val.grad = wavelet_grad[idx]
        for p in m2.parameters():
            if p.grad is not None:
                p.grad.mul_(1.0 / 65536.0)
        opt2.step()
        losses2.append(float(mse2 + reg2))
    # (2) TrainStep
    m1 = copy.deepcopy(m0)
    ts = TrainStep(m1, lr=1e-2, wavelet_regularization=LAM, iters=30000, warmup_steps=0, fp16=True, init_scale=65536.0,
                   update_extra_interval=0, min_wavelet_resolution_to_learn=thr, defer_adam=windowed)
    m1.mean_count = 0
    losses1 = [float(ts.step(t(o), t(d), gt, noises=noise)) for _ in range(3)]
    if windowed:
        assert ts._roi is not None and ts.defer_adam and ts._pending > 0, (ts._roi, ts._pending)
    total1 = sum(losses1) + float(ts.pop_deferred_reg())
    ts.flush_deferred()
    assert ts.frozen_levels == sum(1 for n in (64, 128, 256) if n <= thr and n < Rt) and ts.freeze_ll and ts.freeze_mlp
    np.testing.assert_allclose(total1, sum(losses2), rtol=2e-3)
    p1, p2 = dict(m1.named_parameters()), dict(m2.named_parameters())
    for n, p in init.items():
        a, b = p1[n].detach(), p2[n].detach()
        if not learns(n, p):
            assert torch.equal(a, p) and torch.equal(b, p), n                 # frozen: the same bits in both loops
            continue
        assert not torch.equal(a, p), n
        g2 = p2[n].grad.detach().abs()
        sig = g2 > 1e-2 * g2.max()
        frac = float(((a - b).abs()[sig] > 5e-4).float().mean())
        assert frac < 1e-2, (n, frac)
    # the optimiser's moments of the frozen tensors never moved
    assert float(ts.mlp.m.abs().max()) == 0 and float(ts.ll.m.abs().max()) == 0
    for lvl in range(ts.frozen_levels):
        o_, n_ = ts.coef.offsets[lvl], ts.coef.sizes[lvl]
        assert float(ts.coef.m[o_:o_ + n_].abs().max()) == 0 and float(ts.coef.v[o_:o_ + n_].abs().max()) == 0
