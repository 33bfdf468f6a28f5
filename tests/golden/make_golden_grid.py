"""Generates tests/golden/grid_reference.npz by RUNNING THE REFERENCE's own density-grid upkeep in this container
(never on the GPU box; /root/reference does not travel):

    python tests/golden/make_golden_grid.py

What runs, imported from /root/reference UNMODIFIED (through the stand-ins of make_golden_network.py):
  reconstruction/nerf/renderer.py   NeRFRenderer.mark_untrained_grid (:383-446) and update_extra_state (:448-542): full
                                    refresh twice (the second exercises the EMA max(grid * 0.95, new) against a changed
                                    field), partial refresh (iter_density >= 16), the threshold min(mean_density,
                                    density_thresh) in both regimes, mean_count from a filled step_counter ring with
                                    local_step below and above 16
  aux_libs/raymarching/raymarching.py  morton3D / morton3D_invert / packbits wrappers (over the C oracle's kernels)

Substituted, so that three parties (this run, oracle/grid.py on the CPU, the HIP product on the GPU) can be compared
value for value:
  * the network's density() by the analytic `oracle.grid.blob_density` (+, -, *, clamp only: bit-identical in fp32 on
    CPU and GPU), set on the model INSTANCE -- the reference's methods call self.density(x)['sigma'];
  * torch.rand_like / torch.randint, for the duration of each call, by `oracle.grid.Draws(seed)` (numpy PCG64): the
    fixture stores the seeds, not the draws.
Two grid sizes: H = 32 (every array stored) and the reference's own H = 128 (bitfields, the untrained mask as bits,
every 61st cell of each grid and float64 sums).  grid_size is an attribute the reference's methods read
(`self.grid_size`); for H = 32 the two buffers are re-created at that size.
"""
import contextlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden_network as mgn  # noqa: E402
from oracle import grid as ogrid  # noqa: E402

BOUND = 1.5
# (cx, cy, cz, peak, falloff): a ball of radius 0.6 around the origin, a small one outside [-1,1]^3 (cascade 1 only),
# and a faint wide one that puts most cells just above / below the mean
BLOBS_A = [(0.05, -0.1, 0.02, 40.0, 111.0), (1.2, -0.3, 0.9, 25.0, 400.0), (-0.3, 0.4, 0.2, 0.6, 0.45)]
BLOBS_B = [(0.2, 0.05, -0.15, 30.0, 95.0), (1.2, -0.3, 0.9, 12.0, 400.0), (-0.3, 0.4, 0.2, 0.6, 0.45)]


def poses_and_intrinsic():
    poses = []
    for a in range(5):
        th, ph = 0.5 + 0.3 * a, 1.1 * a
        eye = 2.6 * np.array([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)])
        fwd = -eye / np.linalg.norm(eye)
        right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        pose = np.eye(4, dtype=np.float32)
        # NGP convention of the reference's provider: camera looks along +z of its frame (cam z > 0 is in front, :432)
        pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, up, fwd, eye
        poses.append(pose)
    return np.stack(poses).astype(np.float32), np.array([55.0, 55.0, 20.0, 20.0], np.float32)   # narrow: part of the volume unseen


@contextlib.contextmanager
def patched_rng(draws):
    """torch.rand_like / torch.randint of the reference's update_extra_state served from `draws` (in call order)."""
    real_rand_like, real_randint = torch.rand_like, torch.randint

    def rand_like(t, *a, **k):
        return torch.from_numpy(draws.rand(t.shape)).to(t.dtype)

    def randint(lo, hi, size, *a, **k):
        out = torch.from_numpy(draws.randint(lo, hi, size))
        return out.to(k.get("dtype") or torch.int64)
    torch.rand_like, torch.randint = rand_like, randint
    try:
        yield
    finally:
        torch.rand_like, torch.randint = real_rand_like, real_randint


def density_of(blobs):
    def density(x):
        return {"sigma": torch.from_numpy(ogrid.blob_density(x.detach().numpy().astype(np.float32), blobs)), "geo_feat": None}
    return density


def snapshot(model):
    return dict(grid=model.density_grid.numpy().copy(), bitfield=model.density_bitfield.numpy().copy(),
                mean_density=np.float64(model.mean_density), mean_count=np.int64(model.mean_count),
                iter_density=np.int64(model.iter_density), local_step=np.int64(model.local_step))


def scenario(model, H, thresh, out, tag, full):
    """mark_untrained -> full refresh (field A, 5 ring slots) -> full refresh (field B, ring full, local_step 20) ->
    partial refresh with zero jitter -> partial refresh with jitter."""
    model.grid_size = H
    model.density_grid = torch.zeros(model.cascade, H ** 3)
    model.density_bitfield = torch.zeros(model.cascade * H ** 3 // 8, dtype=torch.uint8)
    model.reset_extra_state()
    model.density_thresh = thresh
    poses, intr = poses_and_intrinsic()
    model.mark_untrained_grid(poses, intr)
    untrained = model.density_grid.numpy() == -1
    assert 0 < untrained.sum() < untrained.size
    out[f"{tag}/untrained"] = np.packbits(untrained.reshape(-1))
    ring = np.zeros((16, 2), np.int32)
    ring[:, 0] = 100000 + 7919 * np.arange(16)
    ring[:, 1] = 60000
    steps = [("full0", BLOBS_A, 5, 11, False, None), ("full1", BLOBS_B, 20, 12, False, None),
             ("part0", BLOBS_A, 3, 13, True, 16), ("part1", BLOBS_B, 16, 14, False, None)]
    for name, blobs, local_step, seed, zero_noise, set_iter in steps:
        model.step_counter.copy_(torch.from_numpy(ring))
        model.local_step = local_step
        if set_iter is not None:
            model.iter_density = set_iter
        model.density = density_of(blobs)
        with patched_rng(ogrid.Draws(seed, zero_noise)):
            model.update_extra_state()
        snap = snapshot(model)
        occ = (snap["grid"] > 0).sum(1)
        print(f"{tag}/{name}: mean_density {snap['mean_density']:.6f} thresh {min(snap['mean_density'], thresh):.6f} "
              f"occupied cells {occ.tolist()} bits set {int(np.unpackbits(snap['bitfield']).sum())} "
              f"mean_count {int(snap['mean_count'])} iter {int(snap['iter_density'])}")
        out[f"{tag}/{name}/seed"] = np.array([seed, int(zero_noise), local_step], np.int64)
        out[f"{tag}/{name}/bitfield"] = snap["bitfield"]
        for k in ("mean_density", "mean_count", "iter_density", "local_step"):
            out[f"{tag}/{name}/{k}"] = snap[k]
        if full:
            out[f"{tag}/{name}/grid"] = snap["grid"]
        else:
            out[f"{tag}/{name}/grid_every61"] = snap["grid"][:, ::61].copy()
            out[f"{tag}/{name}/grid_sum"] = snap["grid"].astype(np.float64).sum(1)
            out[f"{tag}/{name}/grid_abs_sum"] = np.abs(snap["grid"].astype(np.float64)).sum(1)
    out[f"{tag}/ring"] = ring
    out[f"{tag}/cfg"] = np.array([H, model.cascade], np.int64)
    out[f"{tag}/cfg_f"] = np.array([BOUND, thresh, model.density_scale], np.float64)


def main():
    NeRFNetwork, U, get_params, rm_native = mgn.import_reference(check_adjoint=False)
    torch.manual_seed(0)
    sys.argv = ["main_nerf.py", "--path", "/nonexistent", "--workspace", "/tmp/_tnl_golden_ws", "--cuda_ray", "--bound",
                str(BOUND), "--scale", "1", "--dt_gamma", "0", "--triplane_wavelet", "--triplane_channels", "4",
                "--triplane_wavelet_levels", "2", "--triplane_resolution", "16", "--ckpt", "scratch"]
    opt = get_params()
    for k, v in list(vars(opt).items()):
        if isinstance(v, list) and len(v) == 1:
            setattr(opt, k, v[0])
    model = NeRFNetwork(encoding="triplane_wavelet", bound=opt.bound, cuda_ray=True, density_scale=opt.density_scale,
                        min_near=opt.min_near, density_thresh=opt.density_thresh, bg_radius=opt.bg_radius,
                        **{k: vars(opt)[k] for k in mgn.MODEL_KEYS})
    import inspect
    assert inspect.getsourcefile(inspect.unwrap(type(model).update_extra_state)).startswith("/root/reference")
    assert inspect.getsourcefile(inspect.unwrap(type(model).mark_untrained_grid)).startswith("/root/reference")
    out = {}
    poses, intr = poses_and_intrinsic()
    out["poses"], out["intrinsic"] = poses, intr
    out["blobs_a"], out["blobs_b"] = np.array(BLOBS_A, np.float64), np.array(BLOBS_B, np.float64)
    scenario(model, 32, 10.0, out, "g32", full=True)         # threshold = mean_density (opt.density_thresh = 10, main_nerf)
    scenario(model, 128, 0.05, out, "g128", full=False)      # threshold = density_thresh (< mean)
    path = os.path.join(HERE, "grid_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
