"""End-to-end run without a dataset: the README's two-stage "small" recipe (reference README.md:46) scaled down, on the
analytic sphere scene -- device ray pool, fused training step, stage hand-off through a `latest_model` checkpoint,
PSNR on held-out views.  Needs an MI355X:

    python examples/train_synthetic.py [workspace]

The structure mirrors reconstruction/main_nerf.py:168-205 (one model + Trainer per resolution stage).
"""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from trinerflet_amd import synthetic  # noqa: E402
from trinerflet_amd.nerf.network import NeRFNetwork  # noqa: E402
from trinerflet_amd.raypool import RayPool  # noqa: E402
from trinerflet_amd.trainer import train_stages  # noqa: E402


def main():
    workspace = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix="trinerflet_")
    dev = torch.device("cuda:0")
    poses, intr, images = synthetic.sphere_dataset(n_cams=24, H=128, W=128, seed=0)
    train = RayPool(poses[4:], intr, 128, 128, images[4:], device=dev)
    valid = RayPool(poses[:4], intr, 128, 128, images[:4], device=dev)

    def make_model(stage):
        torch.manual_seed(0)
        return NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, density_thresh=10, hidden_dim=64,
                           hidden_dim_color=64, triplane_channels=16, triplane_resolution=stage["triplane_resolution"],
                           triplane_wavelet_levels=stage["triplane_wavelet_levels"], wavelet_type="bior6.8").to(dev)

    stages = [dict(triplane_resolution=256, triplane_wavelet_levels=4, iters=300, num_rays=8192, warmup_steps=0),
              dict(triplane_resolution=512, triplane_wavelet_levels=8, iters=600, num_rays=16384, warmup_steps=50)]
    trainer = train_stages(make_model, lambda s: (train, valid), stages, workspace, name="sphere", lr=1e-2,
                           wavelet_regularization=0.2, background_color=0.0, fast_training=True, mute=False)
    print("held-out views:", trainer.evaluate_one_epoch(valid))
    print("checkpoints in", os.path.join(workspace, "checkpoints"))


if __name__ == "__main__":
    main()
