"""Generates tests/golden/trainer_reference.npz by RUNNING THE REFERENCE's own Python in this container
(never on the GPU box; /root/reference does not travel):

    python tests/golden/make_golden_trainer.py

What runs, imported from /root/reference/reconstruction/nerf/utils.py unmodified (its unrelated top-level imports
cv2, tensorboardX, mcubes, lpips, ... are absent here and are satisfied with empty placeholder modules; none of
them is touched by the functions called below):
  get_rays(poses, intrinsics, H, W, -1)          utils.py:65-149   -> rays/{poses,intrinsics,H,W,rays_o,rays_d}
  decay_function(iter, opt)                      utils.py:55-62    -> lr/{iters,warmup,it,factor}
  PSNRMeter.update / measure                     utils.py:245-282  -> psnr/{pred,truth,value}
  shuffle_data / select_batch shapes             utils.py:228-243  -> batch/{...} (shape contract only)
  sample_pdf(bins, weights, n, det=True)         renderer.py:18-55 -> pdf/{bins,weights,samples}
and the torch objects whose state_dict layout a checkpoint carries (main_nerf.py:119-129, utils.py:1390-1412):
  torch.optim.Adam(model.get_params(lr)) / LambdaLR / GradScaler -> ckpt/{optimizer_keys, scheduler_keys, scaler_keys}
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/reconstruction"


def _placeholder(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def main():
    for name in ["imageio", "tensorboardX", "cv2", "trimesh", "mcubes", "lpips", "torch_ema", "torchmetrics",
                 "torchmetrics.functional", "torchvision", "matplotlib", "matplotlib.pyplot", "raymarching"]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _placeholder(name)
    sys.modules["torch_ema"].ExponentialMovingAverage = object
    sys.modules["torchmetrics.functional"].structural_similarity_index_measure = None
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "..", "aux_libs"))
    from nerf import utils as U  # the reference's module

    out = {}
    # ---- get_rays
    rng = np.random.default_rng(0)
    B, H, W = 3, 20, 24
    poses = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    for b in range(B):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        poses[b, :3, :3] = q.astype(np.float32)
        poses[b, :3, 3] = rng.standard_normal(3).astype(np.float32) * 2
    intr = np.array([31.5, 29.25, 12.0, 10.0], np.float32)
    r = U.get_rays(torch.from_numpy(poses), intr, H, W, -1)
    out.update({"rays/poses": poses, "rays/intrinsics": intr, "rays/HW": np.array([H, W]),
                "rays/rays_o": r["rays_o"].numpy(), "rays/rays_d": r["rays_d"].numpy(),
                "rays/inds": r["inds"].numpy()})
    # ---- decay_function
    its, facs, cfg = [], [], []
    for iters, warm in [(1000, 0), (2000, 100), (6000, 400)]:
        opt = types.SimpleNamespace(iters=iters, warmup_steps=warm, accumelate_steps=1, sched_base=0.1,
                                    warmup_factor=1e-3, sched_exp=2.5)
        for it in [0, 1, warm // 2, max(warm - 1, 0), warm, warm + 1, iters // 2, iters, iters + warm, 2 * iters]:
            its.append(it); cfg.append((iters, warm)); facs.append(U.decay_function(it, opt))
    out.update({"lr/it": np.array(its), "lr/cfg": np.array(cfg), "lr/factor": np.array(facs, np.float64)})
    # ---- PSNRMeter
    pred = rng.random((2, 8, 9, 3)).astype(np.float32)
    truth = rng.random((2, 8, 9, 3)).astype(np.float32)
    m = U.PSNRMeter()
    m.update(torch.from_numpy(pred[:1]), torch.from_numpy(truth[:1]))
    m.update(torch.from_numpy(pred[1:]), torch.from_numpy(truth[1:]))
    out.update({"psnr/pred": pred, "psnr/truth": truth, "psnr/value": np.array(m.measure(), np.float64)})
    # ---- batch selection contract
    data = {"rays_o": torch.arange(2 * 5 * 3, dtype=torch.float32).view(2, 5, 3),
            "images": torch.arange(2 * 5 * 4, dtype=torch.float32).view(2, 5, 4)}
    torch.manual_seed(0)
    sh = U.shuffle_data(data)
    b1 = U.select_batch(sh, 1, 4, torch.device("cpu"))
    out.update({"batch/shuffled_rows": np.array(sh["rays_o"].shape), "batch/sel_shape": np.array(b1["rays_o"].shape),
                "batch/last_shape": np.array(U.select_batch(sh, 2, 4, torch.device("cpu"))["rays_o"].shape),
                "batch/is_perm": np.array(sorted(sh["rays_o"][:, 0].tolist()) == data["rays_o"].view(-1, 3)[:, 0].tolist())})
    # ---- sample_pdf (hierarchical resampling of NeRFRenderer.run)
    from nerf import renderer as RR
    bins = torch.sort(torch.rand(7, 33, generator=torch.Generator().manual_seed(4)) * 3 + 0.5, dim=-1).values
    weights = torch.rand(7, 32, generator=torch.Generator().manual_seed(5))
    weights[2] = 0                                   # a ray that saw nothing
    weights[3, :30] = 0                              # all the mass in the last bins
    samples = RR.sample_pdf(bins, weights, 16, det=True)
    out.update({"pdf/bins": bins.numpy(), "pdf/weights": weights.numpy(), "pdf/samples": samples.numpy()})
    # ---- checkpoint component layouts (torch objects the reference saves)
    p = [torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2, 2))]
    optim_ = torch.optim.Adam([{"params": [p[0]], "lr": 1e-2}, {"params": [p[1]], "lr": 1e-2}], betas=(0.9, 0.99),
                              eps=1e-15)
    (p[0].sum() + p[1].sum()).backward()
    optim_.step()
    sched = torch.optim.lr_scheduler.LambdaLR(optim_, lambda it: 1.0)
    sd = optim_.state_dict()
    out["ckpt/optimizer_state_keys"] = np.array(sorted(sd["state"][0].keys()))
    out["ckpt/optimizer_group_keys"] = np.array(sorted(sd["param_groups"][0].keys()))
    out["ckpt/scheduler_keys"] = np.array(sorted(sched.state_dict().keys()))
    np.savez_compressed(os.path.join(HERE, "trainer_reference.npz"), **out)
    print("wrote trainer_reference.npz:", {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
