"""Experiment: distribution of the plane gradient's tile-list lengths at the base workload (is the tile reduction bound
by its hottest tiles?).  GPU box: PYTHONPATH=. python tools/exp_tile_lists.py"""
import ctypes as C

import numpy as np
import torch
from trinerflet_amd import _lib as L, raymarching, synthetic
from trinerflet_amd.nerf import field as F_

dev = torch.device("cuda:0")
R, bound = 2048, 1.5
bf = torch.from_numpy(synthetic.sphere_bitfield(128, 2, bound, 0.8, 0.0)).to(dev)
o, d = synthetic.training_rays(60000, n_cams=100, seed=0)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
aabb = torch.tensor([-bound] * 3 + [bound] * 3, dtype=torch.float32, device=dev)
nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.2)
counter = torch.zeros(2, dtype=torch.int32, device=dev)
nz = torch.rand(60000, device=dev)
x, dd, dl, rr = raymarching.march_rays_train(o, d, bound, bf, 2, 128, nears, fars, counter, -1, True, 128, False, 0, 1024, nz)
M = int(counter[0])
x = x[:M].contiguous()
ws = F_.plane_grad_sort(x, bound, R)
lay = (C.c_int64 * 5)()
L.check(L.lib().tnl_plane_grad_sort_layout(L.u32(M), L.u32(R), lay), "layout")
nb, off_idx, ent_idx, subs, pos_idx = (int(v) for v in lay)
offsets = ws.view(torch.int32)[off_idx:off_idx + nb + 1][::subs].cpu().numpy().astype(np.int64)
lens = np.diff(offsets)
nz_ = lens[lens > 0]
print(f"samples {M}, entries {offsets[-1]}, tiles {lens.size}, non-empty {nz_.size}")
print("list length: mean of non-empty %.0f, median %.0f, p90 %.0f, p99 %.0f, max %d" %
      (nz_.mean(), np.median(nz_), np.percentile(nz_, 90), np.percentile(nz_, 99), nz_.max()))
srt = np.sort(nz_)[::-1]
print("the 16 longest:", srt[:16].tolist())
print("chunks of 256: total %d, in the longest tile %d" % (int(np.ceil(nz_ / 256).sum()), int(np.ceil(srt[0] / 256))))
