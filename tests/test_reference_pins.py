"""The CPU oracle pinned against the reference's OWN Python, run in the build container and frozen in
tests/golden/network_reference.npz (generator: tests/golden/make_golden_network.py -- the reference's NeRFNetwork,
NeRFRenderer.run / run_cuda, Trainer.train_step, raymarching.py wrappers, TriPlaneVolume, all imported unmodified):

  F-MLP   oracle/field.py::field                 == NeRFNetwork.forward / density / color     (network.py:118-214)
  planes  oracle C build_planes / torch restatement == TriPlaneVolume.get_planes (through real PyWavelets)
  F-RUN   oracle/torch_baseline.py::render_run   == NeRFRenderer.run                          (renderer.py:126-254)
  F-STEP  oracle step (C march/composite + torch field + torch Adam) == Trainer.train_step + backward + Adam
          (utils.py:532-679,1134-1175; main_nerf.py:119,129), two iterations, incl. the sample-budget rule
  glue    image + (1 - ws) bg, depth normalisation of run_cuda's training branch             (renderer.py:317-318)
  F-INFER the eval loop policy of run_cuda                                                    (renderer.py:324-374)

The CUDA kernels themselves (march / composite / SH) appear in those fixtures through the oracle's restatement, so
what these tests pin is every piece of reference Python around them; the GPU tests then compare the HIP path with the
same file (tests/test_reference_fixtures_gpu.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import cref, field as ofield, torch_baseline as tb

NAMES = ["W0", "W1", "W2", "W3", "W4"]


@pytest.fixture(scope="module")
def ref(golden_dir):
    return np.load(os.path.join(golden_dir, "network_reference.npz"))


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _cfg(ref):
    C, R, scale, H, N, max_steps = (int(v) for v in ref["cfg"])
    bound, lam, bg, lr, min_near, dscale = (float(v) for v in ref["cfg_f"])
    return C, R, scale, H, N, max_steps, bound, lam, bg, lr, min_near


def _params(ref, grad=False):
    ll = torch.from_numpy(ref["param/ll"]).clone().requires_grad_(grad)
    coefs = [torch.from_numpy(ref[f"param/coef{i}"]).clone().requires_grad_(grad) for i in range(2)]
    W = [torch.from_numpy(ref[f"param/{n}"]).clone().requires_grad_(grad) for n in NAMES]
    return ll, coefs, W


def test_planes_match_reference_get_planes(ref):
    ll, coefs, _ = _params(ref)
    want = ref["planes"]
    got_c = cref.build_planes(ll.numpy(), [c.numpy() for c in coefs], "bior6.8")
    got_t = ofield.build_planes_torch(ll, coefs, "bior6.8").numpy()
    scale = np.abs(want).max()
    assert np.abs(got_c - want).max() < 2e-6 * scale and np.abs(got_t - want).max() < 2e-6 * scale


def test_field_matches_reference_network(ref):
    """F-MLP: sigma / rgb, density(), masked color(), and the VJP w.r.t. planes and the five weight matrices."""
    _, _, _, _, _, _, bound, *_ = _cfg(ref)
    _, _, W = _params(ref, grad=True)
    planes = torch.from_numpy(ref["planes"]).clone().requires_grad_(True)
    xyz, dirs = torch.from_numpy(ref["mlp/xyz"]), torch.from_numpy(ref["mlp/dirs"])
    sigma, rgb = ofield.field(planes, xyz, dirs, W, bound)
    np.testing.assert_allclose(sigma.detach().numpy(), ref["mlp/sigma"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(rgb.detach().numpy(), ref["mlp/rgb"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(sigma.detach().numpy(), ref["mlp/density_sigma"], rtol=2e-5, atol=1e-7)
    mask = ref["mlp/mask"]
    np.testing.assert_allclose(np.where(mask[:, None], rgb.detach().numpy(), 0.0), ref["mlp/color_masked"], atol=2e-6)
    grads = torch.autograd.grad([sigma, rgb], [planes] + W,
                                [torch.from_numpy(ref["mlp/cot_sigma"]), torch.from_numpy(ref["mlp/cot_rgb"])])
    assert _rel(grads[0].numpy(), ref["mlp/dplanes"]) < 2e-5
    for k in range(5):
        assert _rel(grads[1 + k].numpy(), ref[f"mlp/dW{k}"]) < 2e-5, k


def test_render_run_matches_reference_run(ref):
    """F-RUN / A14: 64 uniform steps, cumprod compositing, colour where weight > 1e-4, background mix, depth."""
    _, _, _, _, _, _, bound, _, bg, _, min_near = _cfg(ref)
    _, _, W = _params(ref)
    o, d = ref["rays/o"], ref["rays/d"]
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = cref.near_far_from_aabb(o, d, aabb, min_near)
    with torch.no_grad():
        out = tb.render_run(torch.from_numpy(ref["planes"]), W, torch.from_numpy(o), torch.from_numpy(d),
                            torch.from_numpy(nears), torch.from_numpy(fars), bound, num_steps=64, bg=bg, full=True)
    hit = np.isfinite(ref["run64/depth"])            # a ray that misses the box has near = far = FLT_MAX -> nan depth
    assert hit.sum() >= 250 and not hit.all()
    for k in ("image", "weights_sum", "depth"):
        got, want = out[k].numpy(), ref[f"run64/{k}"]
        np.testing.assert_allclose(got[hit], want[hit], rtol=0, atol=3e-6, err_msg=k)
    assert np.isnan(out["depth"].numpy()[~hit]).all()


def test_composite_oracle_equals_reference_run_weights(ref):
    """The C restatement of composite_rays_train_forward (raymarching.cu:501-577) against an output of the REFERENCE's
    own Python: with T_thresh = 0 the kernel's recurrence (alpha = 1 - exp(-sigma dt), w = alpha T, T *= 1 - alpha)
    is the cumprod compositing of NeRFRenderer.run (renderer.py:206-229), so the weights_sum / image / depth the
    reference's run() produced for the 64-step fixture (run64/*, make_golden_network.py) must come out of the oracle
    kernel fed with the same sigma / rgb / dt -- sigma and rgb from the oracle field, itself pinned to the reference's
    NeRFNetwork above."""
    _, _, _, _, _, _, bound, _, bg, _, min_near = _cfg(ref)
    _, _, W = _params(ref)
    o, d = torch.from_numpy(ref["rays/o"]), torch.from_numpy(ref["rays/d"])
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = (torch.from_numpy(a) for a in cref.near_far_from_aabb(o.numpy(), d.numpy(), aabb, min_near))
    hit = np.isfinite(ref["run64/depth"])
    N, S = o.shape[0], 64
    nr, fr = nears.unsqueeze(-1), fars.unsqueeze(-1)
    z = nr + (fr - nr) * torch.linspace(0.0, 1.0, S).unsqueeze(0).expand(N, S)                    # renderer.py:150-151
    xyz = (o.unsqueeze(-2) + d.unsqueeze(-2) * z.unsqueeze(-1)).clamp(-bound, bound)
    dt = torch.cat([z[:, 1:] - z[:, :-1], ((fr - nr) / S).expand(N, 1)], -1)                      # :198-199
    with torch.no_grad():
        sig, rgb = ofield.field(torch.from_numpy(ref["planes"]), xyz.reshape(-1, 3),
                                d.view(N, 1, 3).expand(N, S, 3).reshape(-1, 3), W, bound)
    sig, rgb = sig.view(N, S).numpy(), rgb.view(N, S, 3).numpy()
    # run() evaluates colour only where weight > 1e-4 and leaves zeros elsewhere (:213-219)
    a = 1 - np.exp(-dt.numpy().astype(np.float64) * sig)
    w = a * np.cumprod(np.concatenate([np.ones((N, 1)), 1 - a + 1e-15], 1), 1)[:, :-1]
    rgb = np.where((w > 1e-4)[..., None], rgb, 0).astype(np.float32)
    # kernel inputs: deltas[:,0] = dt of the alpha, deltas[:,1] = t-differences whose running sum is the depth's t (= z)
    tdiff = np.concatenate([z[:, :1].numpy(), (z[:, 1:] - z[:, :-1]).numpy()], 1)
    deltas = np.stack([dt.numpy(), tdiff], -1).reshape(-1, 2).astype(np.float32)
    rays = np.stack([np.arange(N), np.arange(N) * S, np.full(N, S)], 1).astype(np.int32)
    keep = np.nonzero(hit)[0]
    ws, dep, img = cref.composite_rays_train_forward(sig.reshape(-1).astype(np.float32), rgb.reshape(-1, 3), deltas, rays, 0.0)
    image = img + (1 - ws)[:, None] * bg
    depth = (dep - nears.numpy() * ws) / (fars.numpy() - nears.numpy())                           # sum w (z - near) / (far - near)
    np.testing.assert_allclose(ws[keep], ref["run64/weights_sum"][keep], rtol=0, atol=5e-6)
    np.testing.assert_allclose(image[keep], ref["run64/image"][keep], rtol=0, atol=5e-6)
    np.testing.assert_allclose(depth[keep], ref["run64/depth"][keep], rtol=0, atol=2e-5)
    assert float(ws[keep].max()) - float(ws[keep].min()) > 0.1 and len(keep) >= 250


class _OracleComposite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sig, rgb, deltas, rays):
        ws, dep, img = cref.composite_rays_train_forward(sig.numpy(), rgb.numpy(), deltas, rays, 1e-4)
        ctx.save = (sig.numpy().copy(), rgb.numpy().copy(), deltas, rays, ws, img)
        return torch.from_numpy(ws), torch.from_numpy(dep), torch.from_numpy(img)

    @staticmethod
    def backward(ctx, gws, gdep, gimg):
        sig, rgb, deltas, rays, ws, img = ctx.save
        gs, gc = cref.composite_rays_train_backward(gws.numpy(), gimg.numpy(), sig, rgb, deltas, rays, ws, img, 1e-4)
        return torch.from_numpy(gs), torch.from_numpy(gc), None, None


def oracle_render_train(ref, ll, coefs, W, noises, mean_count):
    """The oracle's training render: planes -> near/far -> march (budget rule of raymarching.py:195-231) -> field ->
    composite -> background mix + depth normalisation (renderer.py:257-322)."""
    C, R, scale, H, N, max_steps, bound, lam, bg, lr, min_near = _cfg(ref)
    o, d, bf = ref["rays/o"], ref["rays/d"], ref["bitfield"]
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = cref.near_far_from_aabb(o, d, aabb, min_near)
    M = N * max_steps
    if mean_count > 0:
        M = mean_count + (128 - mean_count % 128)
    xyz, dirs, deltas, rays, counter = cref.march_rays_train(o, d, bound, bf, 2, 128, nears, fars, noises, M, 0.0, max_steps)
    if mean_count <= 0:
        m = int(counter[0])
        m += 128 - m % 128
        xyz, dirs, deltas = xyz[:m], dirs[:m], deltas[:m]
    planes = ofield.build_planes_torch(ll, coefs, "bior6.8")
    sig, rgb = ofield.field(planes, torch.from_numpy(xyz), torch.from_numpy(dirs), W, bound)
    ws, dep, img = _OracleComposite.apply(sig, rgb, deltas, rays)
    image = img + (1 - ws).unsqueeze(-1) * bg
    nr, fr = torch.from_numpy(nears), torch.from_numpy(fars)
    depth = torch.clamp(dep - nr, min=0) / (fr - nr)
    return image, depth, ws, counter


def test_two_training_steps_match_reference_trainer(ref):
    """F-STEP / A13 + A10: loss, MSE, regulariser, prediction, every gradient and every parameter after Adam."""
    C, R, scale, H, N, max_steps, bound, lam, bg, lr, min_near = _cfg(ref)
    ll, coefs, W = _params(ref, grad=True)
    params = [ll] + coefs + W
    opt = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.99), eps=1e-15)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: ofield.lr_factor(it, 100, 0))
    images = torch.from_numpy(ref["rays/images"][0])
    gt = images[:, :3] * images[:, 3:] + bg * (1 - images[:, 3:])                  # utils.py:574-577
    keys = ["g_ll", "g_coef0", "g_coef1"] + [f"g_{n}" for n in NAMES]
    after = ["ll_after", "coef0_after", "coef1_after"] + [f"{n}_after" for n in NAMES]
    for it in range(2):
        mean_count = int(ref[f"step{it}/mean_count"])
        opt.zero_grad()
        image, depth, ws, counter = oracle_render_train(ref, ll, coefs, W, ref[f"step{it}/noises"], mean_count)
        assert np.array_equal(counter, ref[f"step{it}/counter"])                    # exact sample count
        np.testing.assert_allclose(gt.numpy(), ref[f"step{it}/gt"], atol=1e-7)
        np.testing.assert_allclose(image.detach().numpy(), ref[f"step{it}/pred"], atol=2e-6)
        mse = ((image - gt) ** 2).mean()
        reg = ofield.wavelet_reg(coefs, lam)
        assert abs(float(mse) - float(ref[f"step{it}/mse"])) < 1e-6 * float(mse)
        assert abs(float(reg) - float(ref[f"step{it}/wavelet_reg"])) < 1e-6 * float(reg)
        assert abs(float(mse + reg) - float(ref[f"step{it}/loss"])) < 1e-6 * float(mse + reg)
        assert abs(opt.param_groups[0]["lr"] - float(ref[f"step{it}/lr"])) < 1e-12
        (mse + reg).backward()
        for p, k in zip(params, keys):
            assert _rel(p.grad.numpy(), ref[f"step{it}/{k}"]) < 5e-5, (it, k)
        opt.step()
        sched.step()
        for p, k, gk in zip(params, after, keys):
            # Adam's early steps are sign-like (eps = 1e-15): compare where the gradient is not rounding noise
            g = ref[f"step{it}/{gk}"]
            sigm = np.abs(g) > 1e-4 * np.abs(g).max()
            diff = np.abs(p.detach().numpy() - ref[f"step{it}/{k}"])
            assert (diff[sigm] > 2e-5).mean() < 2e-3, (it, k, float((diff[sigm] > 2e-5).mean()))
            p.data.copy_(torch.from_numpy(ref[f"step{it}/{k}"]))      # continue from the reference's parameters
            # (torch Adam's state stays consistent: same gradients to 5e-5)


def test_training_branch_glue_matches_reference_run_cuda(ref):
    """A10: image + (1 - ws) * bg and depth = clamp(depth - near, 0) / (far - near) (renderer.py:317-318), on the
    parameters after the two steps, under the sample budget."""
    ll = torch.from_numpy(ref["step1/ll_after"])
    coefs = [torch.from_numpy(ref[f"step1/coef{i}_after"]) for i in range(2)]
    W = [torch.from_numpy(ref[f"step1/{n}_after"]) for n in NAMES]
    with torch.no_grad():
        image, depth, ws, _ = oracle_render_train(ref, ll, coefs, W, ref["glue/noises"], int(ref["glue/mean_count"]))
    hit = np.isfinite(ref["glue/depth"])
    assert not hit.all()
    np.testing.assert_allclose(image.numpy(), ref["glue/image"], atol=2e-6)
    np.testing.assert_allclose(ws.numpy(), ref["glue/weights_sum"], atol=2e-6)
    np.testing.assert_allclose(depth.numpy()[hit], ref["glue/depth"][hit], atol=2e-6)
    assert np.isnan(depth.numpy()[~hit]).all()


def oracle_infer_loop(o, d, nears, fars, bf, bound, max_steps, field_fn, T_thresh=1e-4):
    """run_cuda eval branch (renderer.py:324-374) over the C oracle's march_rays / composite_rays."""
    N = o.shape[0]
    ws, dep, img = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    alive = np.arange(N, dtype=np.int32)
    rt = nears.copy()
    step, hist = 0, []
    while step < max_steps:
        n_alive = alive.shape[0]
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        x, dd, dl = cref.march_rays(n_alive, n_step, alive, rt, o, d, bound, bf, 2, 128, nears, fars,
                                    np.zeros(n_alive, np.float32), 128, 0.0, max_steps)
        s, c = field_fn(x, dd)
        cref.composite_rays(n_alive, n_step, alive, rt, s, c, dl, ws, dep, img, T_thresh)
        alive = alive[alive >= 0]
        hist.append(alive.shape[0])
        step += n_step
    return ws, dep, img, hist


def test_inference_loop_matches_reference_eval_branch(ref):
    """F-INFER / A11: survivors per iteration exact, image / depth / weights to fp32 rounding."""
    C, R, scale, H, N, max_steps, bound, lam, bg, lr, min_near = _cfg(ref)
    _, _, W = _params(ref)
    planes = torch.from_numpy(ref["planes"])
    o, d = ref["rays/o"], ref["rays/d"]
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = cref.near_far_from_aabb(o, d, aabb, min_near)

    def field_fn(x, dd):
        with torch.no_grad():
            s, c = ofield.field(planes, torch.from_numpy(x), torch.from_numpy(dd), W, bound)
        return s.numpy(), c.numpy()
    ws, dep, img, hist = oracle_infer_loop(o, d, nears, fars, ref["bitfield"], bound, max_steps, field_fn)
    assert hist == ref["infer/n_alive_after"].tolist()
    image = img + (1 - ws)[:, None] * bg
    depth = np.clip(dep - nears, 0, None) / (fars - nears)
    hit = np.isfinite(ref["infer/depth"])
    np.testing.assert_allclose(image, ref["infer/image"], atol=2e-6)
    np.testing.assert_allclose(ws, ref["infer/weights_sum"], atol=2e-6)
    np.testing.assert_allclose(depth[hit], ref["infer/depth"][hit], atol=2e-6)
