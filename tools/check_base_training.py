"""Sanity run at the base size (3 x 32 ch x 2048^2, scale 32): 240 fused steps on the analytic sphere scene with and
without the occupancy window / support chain; prints the loss curves and the held-out PSNR.  GPU box:
    PYTHONPATH=. python tools/check_base_training.py [large]      (large: 48 channels, hidden 128)
The first steps run with an untrained occupancy grid (every ray samples its whole path: ~26 M samples per step), which is
what exercises the > 4 GB buffers."""
import sys
import time
import numpy as np
import torch
from trinerflet_amd import synthetic
from trinerflet_amd.nerf.network import NeRFNetwork
from trinerflet_amd.raypool import RayPool
from trinerflet_amd.trainer import Trainer

LARGE = len(sys.argv) > 1 and sys.argv[1] == "large"
CH, HID = (48, 128) if LARGE else (32, 64)
dev = torch.device("cuda:0")
poses, intr, images = synthetic.sphere_dataset(n_cams=40, H=200, W=200, seed=0)
train = RayPool(poses[4:], intr, 200, 200, images[4:], device=dev)
valid = RayPool(poses[:4], intr, 200, 200, images[:4], device=dev)
for use_roi in (False, True, False, True):   # the first run also pays the one-time initialisations
    torch.manual_seed(0)
    m = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, density_thresh=10, hidden_dim=HID,
                    hidden_dim_color=HID, triplane_channels=CH, triplane_resolution=2048, triplane_wavelet_levels=32,
                    wavelet_type="bior6.8").to(dev)
    tr = Trainer("b", m, lr=1e-2, iters=240, num_rays=60000, wavelet_regularization=0.4, fast_training=True,
                 train_step_kwargs=dict(use_roi=use_roi))
    t0 = time.time()
    tr.train(train, None, max_epochs=10)
    torch.cuda.synchronize()
    dt = time.time() - t0
    ev = tr.evaluate_one_epoch(valid)
    print(f"use_roi={use_roi}: {tr.global_step} steps in {dt:.2f} s ({dt / tr.global_step * 1e3:.2f} ms/step incl. grid refreshes), "
          f"epoch losses {[round(x, 5) for x in tr.stats['loss']]}, held-out PSNR {ev['PSNR']:.2f} dB, "
          f"window {tr.ts._roi}")
    del tr, m
    torch.cuda.empty_cache()
