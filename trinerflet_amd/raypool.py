"""RayPool -- device-resident training / evaluation pixels.

Counterpart of the reference's data path for `Trainer.train_one_epoch2`:
  provider.py:284-339 (NeRFDataset.collate/dataloader: per-image rays via get_rays, images [B,H,W,3|4])
  utils.py:786-789 + :218-226 (concat_data), :228-236 (shuffle_data: CPU randperm over all B*H*W pixels and a
  gather of every tensor -- 2.5 GB per epoch at 100 x 800 x 800), :238-243 (select_batch: slice + H2D per step).
Here the poses and images live on the device once; `batch(k, N)` is ONE kernel launch (csrc/rays.hip) producing
rays_o, rays_d and the background-blended ground truth for positions [k*N, (k+1)*N) of the epoch's permutation,
which is a keyed bijection evaluated per ray instead of a materialised randperm.  The last batch of an epoch is
short, exactly as `select_batch` slices it.
"""
import ctypes as C
import math

import torch

from . import _lib as L


class RayPool:
    def __init__(self, poses, intrinsics, H, W, images=None, device="cuda", store_u8=False, color_space="srgb"):
        """poses [B,4,4] cam2world (NGP frame, provider.py:23-31), intrinsics (fx, fy, cx, cy), images [B,H,W,3|4]
        float in [0,1] (or uint8), or None for a pose-only pool (test-time rendering).  color_space="linear"
        converts the colour channels once at load time (utils.py:50-52,561-562 do it per batch)."""
        self.device = torch.device(device)
        self.poses = torch.as_tensor(poses, dtype=torch.float32).to(self.device).contiguous()
        assert self.poses.dim() == 3 and self.poses.shape[1:] == (4, 4)
        self.B, self.H, self.W = int(self.poses.shape[0]), int(H), int(W)
        self.intrinsics = [float(v) for v in intrinsics]
        self._intr = (C.c_float * 4)(*self.intrinsics)
        self.images = None
        self.channels = 0
        self.u8 = False
        if images is not None:
            images = torch.as_tensor(images)
            assert images.shape[:3] == (self.B, self.H, self.W) and images.shape[3] in (3, 4)
            if color_space == "linear":
                images = images.to(torch.float32) / (255.0 if images.dtype == torch.uint8 else 1.0)
                rgb = images[..., :3]
                images = torch.cat([torch.where(rgb < 0.04045, rgb / 12.92, ((rgb + 0.055) / 1.055) ** 2.4),
                                    images[..., 3:]], -1)
            if images.dtype == torch.uint8:
                self.u8 = True
            elif store_u8:
                images = (images.to(torch.float32) * 255.0).round().clamp(0, 255).to(torch.uint8)
                self.u8 = True
            else:
                images = images.to(torch.float32)
            self.images = images.to(self.device).contiguous()
            self.channels = int(images.shape[3])
        self.total = self.B * self.H * self.W
        self.key = 0
        self.shuffled = False

    # -- epoch handling -------------------------------------------------------------------------------------
    def shuffle(self, seed):
        """New permutation for the coming epoch (shuffle_data, utils.py:228-236)."""
        self.key = (int(seed) * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        self.shuffled = True

    def steps_per_epoch(self, batch_size):
        return math.ceil(self.total / batch_size)          # utils.py:1128

    def has_gt(self):
        return self.images is not None

    # -- batches --------------------------------------------------------------------------------------------
    def _launch(self, pix, first, perm, n, bg_color, bg_rand, want_gt, want_pix):
        L.require_cuda(self.poses, self.images, pix, bg_rand)   # no CPU path: the batch is made by csrc/rays.hip
        lib = L.lib()
        rays_o = torch.empty(n, 3, dtype=torch.float32, device=self.device)
        rays_d = torch.empty(n, 3, dtype=torch.float32, device=self.device)
        gt = torch.empty(n, 3, dtype=torch.float32, device=self.device) if want_gt else None
        pix_out = torch.empty(n, dtype=torch.int64, device=self.device) if want_pix else None
        L.check(lib.tnl_ray_batch(L.ptr(self.poses), self._intr, L.u32(self.B), L.u32(self.H), L.u32(self.W),
                                  L.ptr(self.images), L.i32(self.channels), L.i32(int(self.u8)), L.ptr(pix),
                                  L.u64(first), L.u64(self.total if perm else 0), L.u64(self.key), L.u32(n),
                                  L.f32(bg_color), L.ptr(bg_rand), L.ptr(rays_o), L.ptr(rays_d), L.ptr(gt),
                                  L.ptr(pix_out), L.stream()), "ray_batch")
        return rays_o, rays_d, gt, pix_out

    def batch(self, batch_idx, batch_size, bg_color=0.0, bg_rand=None, return_pixels=False):
        """Rays [k*N, min((k+1)*N, total)) of the current epoch order -> dict(rays_o, rays_d, gt_rgb[, pixels])."""
        first = batch_idx * batch_size
        n = min(batch_size, self.total - first)
        if n <= 0:
            raise IndexError("batch index past the end of the epoch")
        if bg_rand is not None:
            bg_rand = bg_rand.to(self.device, torch.float32).contiguous()
            assert bg_rand.shape == (n, 3)
        o, d, gt, pix = self._launch(None, first, self.shuffled, n, float(bg_color), bg_rand, self.has_gt(),
                                     return_pixels)
        out = {"rays_o": o, "rays_d": d, "gt_rgb": gt}
        if return_pixels:
            out["pixels"] = pix
        return out

    def rays_for_pixels(self, pix, bg_color=0.0):
        pix = pix.to(self.device, torch.int64).contiguous()
        o, d, gt, _ = self._launch(pix, 0, False, pix.numel(), float(bg_color), None, self.has_gt(), False)
        return {"rays_o": o, "rays_d": d, "gt_rgb": gt}

    def image_rays(self, index, bg_color=0.0):
        """All H*W rays of image `index` in raster order (get_rays(..., N=-1)) with its blended ground truth."""
        n = self.H * self.W
        o, d, gt, _ = self._launch(None, index * n, False, n, float(bg_color), None, self.has_gt(), False)
        return {"rays_o": o, "rays_d": d, "gt_rgb": gt, "H": self.H, "W": self.W}
