"""SURVEY.md 8(f) ranks 1-3 on the GPU: device ray pool (csrc/rays.hip) against the reference's get_rays golden and
the oracle; the Trainer loop, its checkpoints (reference layout, resume, stage hand-off) and evaluation."""
import os

import numpy as np
import pytest
import torch

from oracle import cref
from trinerflet_amd import synthetic

pytestmark = pytest.mark.gpu


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, "trainer_reference.npz"))


def test_image_rays_match_reference_get_rays(cuda, golden_dir):
    from trinerflet_amd.raypool import RayPool
    g = _golden(golden_dir)
    H, W = (int(v) for v in g["rays/HW"])
    pool = RayPool(g["rays/poses"], g["rays/intrinsics"], H, W, device=cuda)
    for b in range(pool.B):
        r = pool.image_rays(b)
        assert torch.equal(r["rays_o"].cpu(), torch.from_numpy(g["rays/rays_o"][b]))
        np.testing.assert_allclose(r["rays_d"].cpu().numpy(), g["rays/rays_d"][b], rtol=0, atol=2e-7)
        assert r["gt_rgb"] is None


@pytest.mark.parametrize("u8", [False, True])
def test_shuffled_batches_vs_oracle(cuda, u8):
    from trinerflet_amd.raypool import RayPool
    rng = np.random.default_rng(3)
    B, H, W = 5, 33, 47
    poses = synthetic.hemisphere_poses(B, seed=2)
    intr = (40.0, 41.5, W / 2, H / 2)
    images = rng.random((B, H, W, 4)).astype(np.float32)
    if u8:
        images = (images * 255).round().astype(np.uint8)
    pool = RayPool(poses, intr, H, W, images, device=cuda)
    pool.shuffle(seed=11)
    N = 1000
    seen = []
    for k in range(pool.steps_per_epoch(N)):
        bg = torch.rand(min(N, pool.total - k * N), 3, device=cuda) if k == 1 else None
        out = pool.batch(k, N, bg_color=0.25, bg_rand=bg, return_pixels=True)
        pix = out["pixels"].cpu().numpy()
        n = pix.shape[0]
        assert n == min(N, pool.total - k * N)                       # select_batch: the last batch is short
        want = cref.permute_index(np.arange(k * N, k * N + n), pool.total, pool.key)
        assert np.array_equal(pix, want)
        o, d = cref.get_rays(poses, np.array(intr, np.float32), H, W, pix)
        assert np.array_equal(out["rays_o"].cpu().numpy(), o)
        np.testing.assert_allclose(out["rays_d"].cpu().numpy(), d, rtol=0, atol=2e-7)
        im = images.reshape(-1, 4)[pix].astype(np.float32) / (255.0 if u8 else 1.0)
        bgv = bg.cpu().numpy() if bg is not None else np.float32(0.25)
        gt = im[:, :3] * im[:, 3:] + bgv * (1 - im[:, 3:])          # utils.py:576
        np.testing.assert_allclose(out["gt_rgb"].cpu().numpy(), gt, rtol=0, atol=1e-6)
        seen.append(pix)
    seen = np.concatenate(seen)
    assert np.array_equal(np.sort(seen), np.arange(pool.total))     # one epoch = every pixel exactly once
    pool.shuffle(seed=12)
    assert not np.array_equal(pool.batch(0, N, return_pixels=True)["pixels"].cpu().numpy(), seen[:N])
    # three-channel images pass through unblended
    pool3 = RayPool(poses, intr, H, W, images[..., :3], device=cuda)
    out = pool3.batch(0, 64, bg_color=0.9, return_pixels=True)
    ref = images[..., :3].reshape(-1, 3)[out["pixels"].cpu().numpy()].astype(np.float32) / (255.0 if u8 else 1.0)
    np.testing.assert_allclose(out["gt_rgb"].cpu().numpy(), ref, rtol=0, atol=1e-6)


def _model(dev, R=128, levels=2):
    from trinerflet_amd.nerf.network import NeRFNetwork
    torch.manual_seed(0)
    return NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, density_thresh=10, hidden_dim=64,
                       hidden_dim_color=64, triplane_channels=16, triplane_resolution=R,
                       triplane_wavelet_levels=levels, wavelet_type="bior6.8").to(dev)


def _pools(dev, n_cams=10, hw=64):
    from trinerflet_amd.raypool import RayPool
    poses, intr, images = synthetic.sphere_dataset(n_cams, hw, hw, seed=1)
    train = RayPool(poses[2:], intr, hw, hw, images[2:], device=dev)
    valid = RayPool(poses[:2], intr, hw, hw, images[:2], device=dev)
    return train, valid


def test_trainer_epochs_checkpoint_resume_and_eval(cuda, tmp_path):
    from trinerflet_amd.trainer import Trainer
    train, valid = _pools(cuda)
    kw = dict(lr=1e-2, iters=400, warmup_steps=0, num_rays=2048, wavelet_regularization=0.05, fast_training=True)
    tr = Trainer("t", _model(cuda), workspace=str(tmp_path), use_checkpoint="scratch", **kw)
    assert train.steps_per_epoch(2048) == 16
    before = tr.evaluate_one_epoch(valid)
    tr.train(train, valid, max_epochs=8)
    after = tr.evaluate_one_epoch(valid)
    assert tr.global_step == 8 * 16 and tr.epoch == 8
    assert tr.stats["loss"][-1] < 0.3 * tr.stats["loss"][0]
    assert after["PSNR"] > before["PSNR"] + 5.0 and after["PSNR"] > 18.0, (before, after)
    # evaluate == PSNRMeter over render_image
    pred, _, gt = tr.render_image(valid, 0)
    pred1, _, gt1 = tr.render_image(valid, 1)
    manual = np.mean([-10 * np.log10(float(((p - g) ** 2).mean())) for p, g in ((pred, gt), (pred1, gt1))])
    assert abs(manual - after["PSNR"]) < 1e-3

    # ---- the checkpoint is the reference's dictionary and loads into plain torch objects
    path = os.path.join(str(tmp_path), "checkpoints", "t_ep0008.pth")
    assert os.path.exists(path)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert {"epoch", "global_step", "stats", "mean_count", "mean_density", "model", "optimizer", "lr_scheduler",
            "scaler"} <= set(ck)
    fresh = _model(cuda)
    fresh.load_state_dict(ck["model"], strict=True)
    opt = torch.optim.Adam(fresh.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    opt.load_state_dict(ck["optimizer"])
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 1.0)
    sched.load_state_dict(ck["lr_scheduler"])
    assert sched.last_epoch == tr.global_step
    torch.cuda.amp.GradScaler().load_state_dict(ck["scaler"])
    names = [n for n, _ in fresh.named_parameters()]
    assert names[0] == "encoder.planes_features" and names[1] == "encoder.planes_features_wavelet_coefs.0"
    st = opt.state[fresh.encoder.planes_features]
    assert float(st["step"]) == float(tr.ts.opt_steps) and float(st["step"]) <= tr.global_step
    assert torch.equal(st["exp_avg"].reshape(-1), tr.ts.ll.m[:st["exp_avg"].numel()])

    # ---- resume: a new Trainer with --ckpt latest continues where the first one stands
    m2 = _model(cuda)
    tr2 = Trainer("t", m2, workspace=str(tmp_path), use_checkpoint="latest", **kw)
    assert tr2.global_step == tr.global_step and tr2.epoch == tr.epoch
    assert float(tr2.ts.scale) == float(tr.ts.scale) and float(tr2.ts.opt_steps) == float(tr.ts.opt_steps)
    for (n1, p1), (n2, p2) in zip(tr.model.named_parameters(), m2.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2), n1
    for f1, f2 in ((tr.ts.coef, tr2.ts.coef), (tr.ts.ll, tr2.ts.ll), (tr.ts.mlp, tr2.ts.mlp)):
        assert torch.equal(f1.m, f2.m) and torch.equal(f1.v, f2.v)
    assert torch.equal(tr.model.density_bitfield, m2.density_bitfield)
    assert m2.mean_count == tr.model.mean_count
    tr.epoch += 1
    tr2.epoch += 1
    la, lb = tr.train_one_epoch(train), tr2.train_one_epoch(train)
    assert abs(la - lb) < 0.05 * la, (la, lb)


def test_stage_handoff_keeps_coarse_levels_and_trains_on(cuda, tmp_path):
    """main_nerf.py's resolution stages with --ckpt latest_model: the next stage starts from the previous planes
    (LL and existing levels copied, new finest level zero, occupancy grid kept, fresh optimiser and step count)."""
    from trinerflet_amd.trainer import Trainer, train_stages
    train, valid = _pools(cuda)
    stages = [dict(triplane_resolution=128, triplane_wavelet_levels=2, iters=64, num_rays=2048, warmup_steps=0),
              dict(triplane_resolution=256, triplane_wavelet_levels=4, iters=64, num_rays=2048, warmup_steps=8)]
    made = []

    def make_model(stage):
        made.append(_model(cuda, stage["triplane_resolution"], stage["triplane_wavelet_levels"]))
        return made[-1]

    common = dict(lr=1e-2, wavelet_regularization=0.05, fast_training=True)
    t1 = train_stages(make_model, lambda s: (train, valid), stages[:1], str(tmp_path), name="s", **common)
    p1 = t1.evaluate_one_epoch(valid)["PSNR"]
    m1 = made[0]
    m2 = _model(cuda, 256, 4)
    t2 = Trainer("s", m2, workspace=str(tmp_path), use_checkpoint="latest_model", iters=64, num_rays=2048, **common)
    assert t2.global_step == 0 and float(t2.ts.opt_steps) == 0 and float(t2.ts.coef.m.abs().sum()) == 0
    assert torch.equal(m2.encoder.planes_features, m1.encoder.planes_features)
    assert torch.equal(m2.encoder.planes_features_wavelet_coefs[0], m1.encoder.planes_features_wavelet_coefs[0])
    assert float(m2.encoder.planes_features_wavelet_coefs[1].abs().sum()) == 0
    assert torch.equal(m2.density_grid, m1.density_grid) and m2.mean_count == m1.mean_count
    # same field at twice the resolution (the new level is zero): the hand-off does not lose the fit
    p2_start = t2.evaluate_one_epoch(valid)["PSNR"]
    assert p2_start > p1 - 1.5, (p1, p2_start)
    t2.train(train, valid)
    assert t2.evaluate_one_epoch(valid)["PSNR"] > p2_start - 0.5


def test_per_ray_background_equals_constant_background(cuda):
    """--train_rand_bg hands TrainStep a per-ray [N,3] background (utils.py:568-570): with a constant tensor it must
    reproduce the scalar background_color path; with random colours the epoch loop runs and the loss stays finite."""
    from trinerflet_amd.train import TrainStep
    from trinerflet_amd.trainer import Trainer
    import copy
    train, _ = _pools(cuda, n_cams=4, hw=32)
    batch = train.batch(0, 1024, bg_color=0.3)
    base = _model(cuda)
    base.density_bitfield.copy_(torch.from_numpy(synthetic.sphere_bitfield(128, base.cascade, 1.5, 0.8, 0.0)).to(cuda))
    noise = torch.rand(1024, device=cuda)
    outs = []
    for per_ray in (False, True):
        m = copy.deepcopy(base)
        ts = TrainStep(m, background_color=0.3, update_extra_interval=0)
        m.mean_count = 0
        bg = torch.full((1024, 3), 0.3, device=cuda) if per_ray else None
        loss = ts.step(batch["rays_o"], batch["rays_d"], batch["gt_rgb"], noises=noise, bg_color=bg)
        outs.append((float(loss), ts.last["image"].clone(), m.sigma_net[0].weight.detach().clone()))
    assert abs(outs[0][0] - outs[1][0]) < 1e-7 and torch.equal(outs[0][1], outs[1][1])
    assert torch.allclose(outs[0][2], outs[1][2], atol=1e-6)
    tr = Trainer("rb", _model(cuda), lr=1e-2, iters=50, num_rays=1024, train_rand_bg=True, fast_training=True)
    tr.train(train, None, max_epochs=2)
    assert np.isfinite(tr.stats["loss"]).all() and tr.global_step == 2 * train.steps_per_epoch(1024)


def test_save_triplane_dumps(cuda, tmp_path):
    """Trainer.save_triplane (utils.py:1600-1661): file set, min-max + contrast normalisation, the wavelet pyramid."""
    from trinerflet_amd.trainer import Trainer
    tr = Trainer("t", _model(cuda, R=128, levels=4), workspace=str(tmp_path), use_checkpoint="scratch", lr=1e-2, iters=10,
                 num_rays=512, fast_training=True)
    with torch.no_grad():
        for p in tr.model.encoder.planes_features_wavelet_coefs:
            p.normal_(0, 0.1)
    files = tr.save_triplane(all=True, save_wavelet=True)
    names = sorted(os.path.relpath(f, str(tmp_path)) for f in files)
    assert len([n for n in names if n.startswith("planes/plane_0_")]) == 3 * 16
    assert len([n for n in names if n.startswith("planes/wavelet_features/")]) == 3 * 16
    assert {n.split("/")[1] for n in names if "levels_" in n} == {"levels_0", "levels_1", "levels_2"}
    # one image against the definition: min-max, then clamp(2x - mean, 0, 1), 8-bit
    tr.model.encoder.reset_cahce()
    pl = tr.model.encoder.get_planes()[1, 5].detach().float().cpu()
    x = (pl - pl.min()) / (pl.max() - pl.min())
    want = ((2 * x - x.mean()).clamp(0, 1) * 255).round().numpy().astype(np.uint8)
    raw = open(os.path.join(str(tmp_path), "planes", "plane_0_1_5.pgm"), "rb").read()
    head, data = raw.split(b"\n", 1)
    assert head == b"P5 128 128 255" and np.array_equal(np.frombuffer(data, np.uint8).reshape(128, 128), want)
    w = open(os.path.join(str(tmp_path), "planes", "wavelet_features", "wavelet_features_0_0_0.pgm"), "rb").read()
    assert w.split(b"\n", 1)[0] == b"P5 128 128 255"          # 32 (LL) -> 64 -> 128 pyramid
