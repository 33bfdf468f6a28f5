"""NeRFRenderer -- mirror of reconstruction/nerf/renderer.py (reference): same constructor, buffers
(aabb_train, aabb_infer, density_grid, density_bitfield, step_counter), attributes (cuda_ray, bg_radius,
mean_count, mean_density, iter_density, local_step) and methods render / run / run_cuda /
update_extra_state / mark_untrained_grid / reset_extra_state.

The marching, compositing, compaction and field evaluation are HIP kernels (trinerflet_amd.raymarching,
NeRFNetwork.forward).  The density-grid refresh evaluates cells in Morton order from a cached coordinate
table instead of the reference's five nested Python loops; its sampling noise comes from torch's RNG
like the reference, so it is distribution- not bit-identical.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from .. import raymarching


def sample_pdf(bins, weights, n_samples, det=False):
    """Inverse-transform sampling of a piecewise-constant density (renderer.py:18-55): bins [B,T], weights [B,T-1] ->
    [B,n_samples] positions; det=True takes the midpoints of n_samples equal probability strata."""
    pdf = weights + 1e-5
    pdf = pdf / pdf.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)             # [B,T]
    if det:
        u = torch.linspace(0.5 / n_samples, 1.0 - 0.5 / n_samples, steps=n_samples, device=weights.device)
        u = u.expand(*cdf.shape[:-1], n_samples)
    else:
        u = torch.rand(*cdf.shape[:-1], n_samples, device=weights.device)
    u = u.contiguous()
    hi = torch.searchsorted(cdf, u, right=True)
    lo = (hi - 1).clamp(min=0)
    hi = hi.clamp(max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    b_lo, b_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    span = c_hi - c_lo
    span = torch.where(span < 1e-5, torch.ones_like(span), span)
    return b_lo + (u - c_lo) / span * (b_hi - b_lo)


class NeRFRenderer(nn.Module):
    def __init__(self, bound=1, cuda_ray=False, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=-1,
                 **kwargs):
        super().__init__()
        # reference: renderer.py:62-100
        self.bound = bound
        self.cascade = 1 + math.ceil(math.log2(bound))
        self.grid_size = 128
        self.density_scale = density_scale
        self.min_near = min_near
        self.density_thresh = density_thresh
        self.bg_radius = bg_radius
        # run_cuda -> forward, while the field is evaluated on a marched batch: the march's device counter, and the window
        # of the planes such a batch can touch (NeRFNetwork._occupancy_window; None: whole planes)
        self._march_count = None
        self._march_window = None
        self.use_occupancy_window = True
        self._occ_window_key = None
        self._occ_window = None
        aabb_train = torch.FloatTensor([-bound, -bound, -bound, bound, bound, bound])
        self.register_buffer('aabb_train', aabb_train)
        self.register_buffer('aabb_infer', aabb_train.clone())
        self.cuda_ray = cuda_ray
        if cuda_ray:
            self.register_buffer('density_grid', torch.zeros([self.cascade, self.grid_size ** 3]))
            self.register_buffer('density_bitfield',
                                 torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
            self.mean_density = 0
            self.iter_density = 0
            self.register_buffer('step_counter', torch.zeros(16, 2, dtype=torch.int32))
            self.mean_count = 0
            self.local_step = 0
        self._morton_xyz = None  # [H^3, 3] cell coords (float, in [-1,1]) listed in Morton order

    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def color(self, x, d, mask=None, **kwargs):
        raise NotImplementedError()

    def reset_extra_state(self):
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.mean_density = 0
        self.iter_density = 0
        self.step_counter.zero_()
        self.mean_count = 0
        self.local_step = 0

    # ------------------------------------------------------------------------------------------
    # non-cuda_ray renderer (renderer.py:126-254): uniform steps (+ optional importance resampling), torch
    # compositing.  The CPU-baseline semantics of SURVEY.md A14; no README configuration trains with it.
    # ------------------------------------------------------------------------------------------
    def run(self, rays_o, rays_d, num_steps=128, upsample_steps=128, bg_color=None, perturb=False, **kwargs):
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N, device = rays_o.shape[0], rays_o.device
        aabb = self.aabb_train if self.training else self.aabb_infer
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)
        nears, fars = nears.unsqueeze(-1), fars.unsqueeze(-1)
        z_vals = torch.linspace(0.0, 1.0, num_steps, device=device).unsqueeze(0).expand(N, num_steps)
        z_vals = nears + (fars - nears) * z_vals
        sample_dist = (fars - nears) / num_steps
        if perturb:
            z_vals = z_vals + (torch.rand(z_vals.shape, device=device) - 0.5) * sample_dist
        xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z_vals.unsqueeze(-1)
        xyzs = torch.min(torch.max(xyzs, aabb[:3]), aabb[3:])
        dens = self.density(xyzs.reshape(-1, 3))
        sigma = dens['sigma'].view(N, num_steps)
        geo = dens['geo_feat'].view(N, num_steps, -1)
        if upsample_steps > 0:
            # renderer.py:176-213: resample along each ray where the coarse pass put its weight, then merge by depth
            with torch.no_grad():
                d0 = z_vals[..., 1:] - z_vals[..., :-1]
                d0 = torch.cat([d0, sample_dist * torch.ones_like(d0[..., :1])], dim=-1)
                a0 = 1 - torch.exp(-d0 * self.density_scale * sigma)
                w0 = a0 * torch.cumprod(torch.cat([torch.ones_like(a0[..., :1]), 1 - a0 + 1e-15], dim=-1), dim=-1)[..., :-1]
                z_mid = z_vals[..., :-1] + 0.5 * d0[..., :-1]
                new_z = sample_pdf(z_mid, w0[:, 1:-1], upsample_steps, det=not self.training).detach()
                new_xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * new_z.unsqueeze(-1)
                new_xyzs = torch.min(torch.max(new_xyzs, aabb[:3]), aabb[3:])
            new_dens = self.density(new_xyzs.reshape(-1, 3))
            z_vals, order = torch.sort(torch.cat([z_vals, new_z], dim=1), dim=1)
            xyzs = torch.gather(torch.cat([xyzs, new_xyzs], dim=1), 1, order.unsqueeze(-1).expand(-1, -1, 3))
            sigma = torch.gather(torch.cat([sigma, new_dens['sigma'].view(N, upsample_steps)], dim=1), 1, order)
            geo = torch.cat([geo, new_dens['geo_feat'].view(N, upsample_steps, -1)], dim=1)
            geo = torch.gather(geo, 1, order.unsqueeze(-1).expand(-1, -1, geo.shape[-1]))
        deltas = z_vals[..., 1:] - z_vals[..., :-1]
        deltas = torch.cat([deltas, sample_dist * torch.ones_like(deltas[..., :1])], dim=-1)
        alphas = 1 - torch.exp(-deltas * self.density_scale * sigma)
        alphas_shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
        weights = alphas * torch.cumprod(alphas_shifted, dim=-1)[..., :-1]
        dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
        mask = weights > 1e-4
        rgbs = self.color(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), mask=mask.reshape(-1),
                          geo_feat=geo.reshape(-1, geo.shape[-1]))
        rgbs = rgbs.view(N, -1, 3)
        weights_sum = weights.sum(dim=-1)
        ori_z_vals = ((z_vals - nears) / (fars - nears)).clamp(0, 1)
        depth = torch.sum(weights * ori_z_vals, dim=-1)
        image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)
        if bg_color is None:
            bg_color = 1
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        return {'depth': depth.view(*prefix), 'image': image.view(*prefix, 3), 'weights_sum': weights_sum}

    # ------------------------------------------------------------------------------------------
    # cuda_ray renderer (renderer.py:257-381)
    # ------------------------------------------------------------------------------------------
    def run_cuda(self, rays_o, rays_d, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False,
                 max_steps=1024, T_thresh=1e-4, noises=None, **kwargs):
        # `noises` ([N] in [0,1), optional) is this build's only extra keyword: an explicit perturbation for
        # seeded parity runs; None = torch.rand as in the reference
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N, device = rays_o.shape[0], rays_o.device
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d,
                                                     self.aabb_train if self.training else self.aabb_infer,
                                                     self.min_near)
        # the march stops at the ray's exit from the box of the occupied cells (no sample lies behind it: the same samples
        # to the bit, raymarching.clip_fars); the depth normalisation below keeps the box's far (renderer.py:318)
        fars_aabb = fars
        if getattr(self, "clip_far_to_occupancy", True):
            # (TrainStep, whose march runs beside other kernels on a side stream, measured no gain from the clip; this
            #  in-line march of the module path does: the count pass no longer walks the empty cells behind the object)
            bf = self.density_bitfield
            key = (bf.data_ptr(), bf._version)
            if getattr(self, "_occ_box_key", None) != key:      # the box belongs to a bitfield: rebuilt when that changes
                self._occ_box_cached = raymarching.occupied_box(bf, self.cascade, self.grid_size, self.bound)
                self._occ_box_key = key
            fars = raymarching.clip_fars(rays_o, rays_d, fars, self._occ_box_cached)
        if bg_color is None:
            bg_color = 1
        results = {}
        if self.training:
            counter = self.step_counter[self.local_step % 16]
            counter.zero_()
            self.local_step += 1
            xyzs, dirs, deltas, rays = raymarching.march_rays_train(
                rays_o, rays_d, self.bound, self.density_bitfield, self.cascade, self.grid_size, nears, fars, counter,
                self.mean_count, perturb, 128, force_all_rays, dt_gamma, max_steps, noises)
            # rows past counter[0] are the zero padding up to the sample budget: the fused field skips them (field.py)
            self._march_count = counter
            self._march_window = self._occupancy_window() if hasattr(self, "_occupancy_window") else None
            try:
                sigmas, rgbs = self(xyzs, dirs)
            finally:
                self._march_count = self._march_window = None
            sigmas = self.density_scale * sigmas
            weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh)
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            depth = torch.clamp(depth - nears, min=0) / (fars_aabb - nears)
            image = image.view(*prefix, 3)
            depth = depth.view(*prefix)
            results['weights_sum'] = weights_sum
        elif self._eval_render_mode(kwargs) == "kernel":
            # render_mode="kernel" (the default of the eval branch): one persistent kernel instead of the alive-ray loop
            # (csrc/render.hip).  Same samples per ray, same order, same arithmetic; a ray still alive after max_steps
            # samples stops exactly there (the loop's cap depends on its schedule: max_steps ... max_steps + 7).
            weights_sum, depth, image = self._infer_render_kernel(rays_o, rays_d, nears, fars, dt_gamma, perturb,
                                                                  max_steps, T_thresh)
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            depth = torch.clamp(depth - nears, min=0) / (fars_aabb - nears)
            image = image.view(*prefix, 3)
            depth = depth.view(*prefix)
            weights_sum = weights_sum.view(*prefix)
        elif self._eval_render_mode(kwargs) == "device_loop":
            weights_sum, depth, image = self._infer_device_loop(rays_o, rays_d, nears, fars, dt_gamma, perturb,
                                                                max_steps, T_thresh,
                                                                min_step=kwargs.get("infer_min_step", 1))
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            depth = torch.clamp(depth - nears, min=0) / (fars_aabb - nears)
            image = image.view(*prefix, 3)
            depth = depth.view(*prefix)
            weights_sum = weights_sum.view(*prefix)
        else:
            weights_sum = torch.zeros(N, dtype=torch.float32, device=device)
            depth = torch.zeros(N, dtype=torch.float32, device=device)
            image = torch.zeros(N, 3, dtype=torch.float32, device=device)
            n_alive = N
            rays_alive = torch.arange(n_alive, dtype=torch.int32, device=device)
            rays_t = nears.clone()
            step = 0
            while step < max_steps:
                if n_alive <= 0:
                    break
                n_step = max(min(N // n_alive, 8), 1)
                xyzs, dirs, deltas = raymarching.march_rays(
                    n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, self.bound, self.density_bitfield,
                    self.cascade, self.grid_size, nears, fars, 128, perturb if step == 0 else False, dt_gamma,
                    max_steps)
                sigmas, rgbs = self(xyzs, dirs)
                sigmas = self.density_scale * sigmas
                raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum,
                                           depth, image, T_thresh)
                # device-side ordered compaction (replaces rays_alive[rays_alive >= 0], renderer.py:364);
                # the survivor count is the one host read per iteration the reference also pays (:345)
                rays_alive, n_out = raymarching.compact_rays(rays_alive, n_alive)
                n_alive = int(n_out.item())
                step += n_step
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            depth = torch.clamp(depth - nears, min=0) / (fars_aabb - nears)
            image = image.view(*prefix, 3)
            depth = depth.view(*prefix)
            weights_sum = weights_sum.view(*prefix)
        results['depth'] = depth
        results['image'] = image
        results['weights_sum'] = weights_sum
        return results

    RENDER_MODES = ("kernel", "device_loop", "host_loop")

    def _eval_render_mode(self, kwargs):
        """Which form of the inference branch (renderer.py:324-374) render(..., render_mode=...) runs:
          "kernel"      one persistent kernel (csrc/render.hip) -- the default;
          "device_loop" the reference's alive-ray loop with its sizes on the device (takes infer_min_step);
          "host_loop"   the reference's structure literally: one survivor count read back per iteration.
        The older keywords still select a form when render_mode is absent: fused_render=True/False, device_loop=True/False,
        infer_min_step=n (a knob of the loop: it implies "device_loop").  The two fused forms need a configuration the
        fused field kernel is built for and no autograd; otherwise the host loop runs."""
        mode = kwargs.get("render_mode")
        if mode is None:
            if "fused_render" in kwargs:
                mode = "kernel" if kwargs["fused_render"] else \
                    ("device_loop" if kwargs.get("device_loop", True) else "host_loop")
            elif "device_loop" in kwargs:
                mode = "device_loop" if kwargs["device_loop"] else "host_loop"
            elif "infer_min_step" in kwargs:
                mode = "device_loop"
            else:
                mode = "kernel"
        elif mode not in self.RENDER_MODES:
            raise ValueError(f"render_mode must be one of {self.RENDER_MODES}, got {mode!r}")
        elif mode != "device_loop" and kwargs.get("infer_min_step", 1) != 1:
            raise ValueError("infer_min_step is a knob of render_mode='device_loop'")
        if mode != "host_loop" and (not getattr(self, "_fused_ok", lambda: False)() or torch.is_grad_enabled()):
            mode = "host_loop"
        return mode

    def _infer_render_kernel(self, rays_o, rays_d, nears, fars, dt_gamma, perturb, max_steps, T_thresh):
        """renderer.py:324-374 as ONE launch (tnl_render_rays): march, fused field and compositing per ray inside a
        persistent kernel, no sample buffers.  Same samples, same order and same arithmetic per ray as the loop."""
        lib = L.lib()
        N, dev = rays_o.shape[0], rays_o.device
        weights_sum = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        if N == 0:
            return weights_sum, depth, image
        enc = self.encoder
        tm = enc.get_planes_texel_major()
        packed = self.packed_weights()
        noises = torch.rand(N, dtype=torch.float32, device=dev) if perturb else None
        queue = torch.empty(1, dtype=torch.int32, device=dev)
        L.check(lib.tnl_render_rays(
            L.ptr(tm), L.i32(int(tm.dtype == torch.float16)), L.u32(enc.number_of_features), L.u32(enc.plane_resolution),
            L.u32(self.hidden_dim), L.u32(self.hidden_dim), L.ptr(packed), L.ptr(rays_o), L.ptr(rays_d), L.ptr(nears),
            L.ptr(fars), L.u32(N), L.ptr(self.density_bitfield), L.f32(self.bound), L.f32(dt_gamma), L.u32(max_steps),
            L.u32(self.cascade), L.u32(self.grid_size), L.f32(T_thresh), L.f32(float(self.density_scale)), L.ptr(noises),
            L.ptr(queue), L.ptr(weights_sum), L.ptr(depth), L.ptr(image), L.stream()), "render_rays")
        return weights_sum, depth, image

    def _infer_device_loop(self, rays_o, rays_d, nears, fars, dt_gamma, perturb, max_steps, T_thresh, poll=4,
                           min_step=1):
        """The alive-ray loop of renderer.py:338-372 with its sizes on the device (include/trinerflet_hip.h,
        tnl_infer_plan ...): iterations are enqueued back to back, the host reads the state only every `poll`
        iterations to stop.  Same n_step rule, ray order and arithmetic as the host-driven loop below."""
        lib = L.lib()
        N, dev = rays_o.shape[0], rays_o.device
        weights_sum = torch.zeros(N, dtype=torch.float32, device=dev)
        depth = torch.zeros(N, dtype=torch.float32, device=dev)
        image = torch.zeros(N, 3, dtype=torch.float32, device=dev)
        if N == 0:
            return weights_sum, depth, image
        state = torch.tensor([N, 0, 0, 0], dtype=torch.int32, device=dev)
        alive = [torch.arange(N, dtype=torch.int32, device=dev), torch.empty(N, dtype=torch.int32, device=dev)]
        rays_t = nears.clone()
        # min_step: the reference starts with one sample per ray and iteration (n_step = max(min(N / n_alive, 8), 1));
        # render(..., infer_min_step=8) regroups the same per-ray sample sequences into an eighth of the iterations
        # (the set-up of 640 000 rays is paid ~60 times instead of ~470).  Rays that end before the max_steps cap get
        # identical colours; the default 1 keeps the reference's schedule exactly.
        min_step = max(1, min(int(min_step), 8))
        cap = min_step * N + 128
        xyzs = torch.empty(cap, 3, dtype=torch.float32, device=dev)
        dirs = torch.empty(cap, 3, dtype=torch.float32, device=dev)
        deltas = torch.empty(cap, 2, dtype=torch.float32, device=dev)
        t_scratch = torch.empty(cap, dtype=torch.float32, device=dev)     # the march's record of the samples' t
        cws = torch.empty((N + 255) // 256 + 2, dtype=torch.int32, device=dev)
        noises = torch.rand(N, dtype=torch.float32, device=dev) if perturb else None
        rows = state[3:4]
        packed = self.packed_weights()     # once per render, not once per iteration (one 5-us launch each: 2.6 % of an image)

        def iteration(nz):
            L.check(lib.tnl_infer_plan(L.ptr(state), L.u32(N), L.u32(max_steps), L.u32(min_step), L.stream()),
                    "infer_plan")
            L.check(lib.tnl_march_rays_dev(
                L.ptr(state), L.u32(N), L.ptr(alive[0]), L.ptr(rays_t), L.ptr(rays_o), L.ptr(rays_d),
                L.f32(self.bound), L.f32(dt_gamma), L.u32(max_steps), L.u32(self.cascade), L.u32(self.grid_size),
                L.ptr(self.density_bitfield), L.ptr(fars), L.ptr(xyzs), L.ptr(dirs), L.ptr(deltas),
                L.ptr(nz), L.ptr(t_scratch), L.u32(cap), L.stream()), "march_rays_dev")
            sigmas, rgbs = self.field_rows(xyzs, dirs, rows, packed)
            if self.density_scale != 1:
                sigmas = self.density_scale * sigmas
            L.check(lib.tnl_composite_rays_dev(
                L.ptr(state), L.u32(N), L.f32(T_thresh), L.ptr(alive[0]), L.ptr(rays_t), L.ptr(sigmas),
                L.ptr(rgbs), L.ptr(deltas), L.ptr(weights_sum), L.ptr(depth), L.ptr(image), L.stream()),
                "composite_rays_dev")
            L.check(lib.tnl_compact_rays_dev(L.ptr(state), L.u32(N), L.ptr(alive[0]), L.ptr(alive[1]),
                                             L.ptr(cws), L.stream()), "compact_rays_dev")
            alive.reverse()

        def finished():
            n_alive, _, step, _ = state.tolist()
            return n_alive <= 0 or step >= max_steps

        # (replaying the iterations as a captured hipGraph was tried: 24.9 vs 23.6 ms per 800x800 image -- the loop is
        #  bound by the device work of its early, full-width iterations, not by the host's launch rate)
        it = 0
        while it < max_steps:                      # every iteration advances `step` by at least 1
            for _ in range(poll):
                iteration(noises if it == 0 else None)
                it += 1
            if finished():
                break
        return weights_sum, depth, image

    # ------------------------------------------------------------------------------------------
    # density grid upkeep
    # ------------------------------------------------------------------------------------------
    def _cells(self):
        """Cell coordinates in [-1,1]^3 (2*c/(H-1) - 1, renderer.py:416,475) listed in Morton order."""
        if self._morton_xyz is None or self._morton_xyz.device != self.density_bitfield.device:
            H = self.grid_size
            idx = torch.arange(H ** 3, dtype=torch.int32, device=self.density_bitfield.device)
            coords = raymarching.morton3D_invert(idx)
            self._morton_xyz = self._cell_centres(coords)
        return self._morton_xyz

    def _cell_centres(self, coords):
        """2 * c / (H - 1) - 1 (renderer.py:416,475) with a TRUE division.  torch's GPU kernel for `tensor / python_scalar`
        multiplies by the scalar's reciprocal; torch's CPU kernel divides.  Parity here is with the reference RUN ON CPU
        (the generator of tests/golden/grid_reference.npz, the only way the reference runs in the build container): the
        true division reproduces its cell centres bit for bit.  The reference on CUDA takes the reciprocal shortcut itself,
        so this is one ulp away from THAT run for about one coordinate in nine -- far below the jitter of half a cell
        added right after (INTEGRATION.md A.2 lists the deviation)."""
        den = torch.full((1,), float(self.grid_size - 1), dtype=torch.float32, device=coords.device)
        return 2 * coords.float() / den - 1

    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        # reference: renderer.py:383-446 -- a cell is trainable iff some camera sees its centre
        if not self.cuda_ray:
            return
        if isinstance(poses, np.ndarray):
            poses = torch.from_numpy(poses)
        fx, fy, cx, cy = intrinsic
        dev = self.density_grid.device
        poses = poses.to(dev).float()
        cells = self._cells()
        count = torch.zeros_like(self.density_grid)
        chunk = 64 ** 3
        for cas in range(self.cascade):
            bound = min(2 ** cas, self.bound)
            half_grid_size = bound / self.grid_size
            for c0 in range(0, cells.shape[0], chunk):
                world = (cells[c0:c0 + chunk] * (bound - half_grid_size)).unsqueeze(0)
                for head in range(0, poses.shape[0], S):
                    P = poses[head:head + S]
                    cam = (world - P[:, :3, 3].unsqueeze(1)) @ P[:, :3, :3]
                    mask = (cam[:, :, 2] > 0) \
                        & (torch.abs(cam[:, :, 0]) < cx / fx * cam[:, :, 2] + half_grid_size * 2) \
                        & (torch.abs(cam[:, :, 1]) < cy / fy * cam[:, :, 2] + half_grid_size * 2)
                    count[cas, c0:c0 + chunk] += mask.sum(0)
        self.density_grid[count == 0] = -1

    def _sigma_for_grid(self):
        """The density the grid refresh queries: self.density(x)['sigma'] (renderer.py:487,511).  The stock network's
        sigma-only fused form (network.density_sigma) is taken only when density() is the stock one -- a subclass or an
        instance that overrides density() (custom density, activation, scale) is queried through ITS density()."""
        stock = getattr(type(self), "_stock_density", None)
        fast = getattr(self, "density_sigma", None)
        if fast is not None and stock is not None and "density" not in self.__dict__ \
                and getattr(type(self).density, "__func__", type(self).density) is stock:
            return fast
        return lambda p_: self.density(p_)['sigma']

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128, shard=None, draws=None):
        # reference: renderer.py:448-542.  shard=(rank, world, gather): multi-GPU refresh: every
        # rank evaluates 1/world of the cells (full refresh: a contiguous block of the Morton-ordered cell list;
        # partial refresh: 1/world of the uniform and of the occupied picks) and `gather` (rank-ordered all-gather
        # along dim 0) completes the candidate grid identically on every rank.
        # draws: explicit random draws for seeded parity runs (None = torch's RNG as in the reference), a dict with
        #   "noise":  per cascade the in-cell jitter in [0,1) (torch.rand_like of renderer.py:484,509), [cells,3] in the
        #             order the cells are evaluated here (full refresh: Morton order; partial: uniform picks, then occupied),
        #   "coords": per cascade the uniform picks [N,3] (torch.randint of :494),
        #   "occ_k":  per cascade the ranks [N] of the occupied picks (torch.randint(0, n_occupied, [N]) of :499).
        if not self.cuda_ray:
            return
        dev = self.density_bitfield.device
        H = self.grid_size
        rank, world, gather = shard if shard is not None else (0, 1, None)
        tmp_grid = -torch.ones_like(self.density_grid)
        sigma_of = self._sigma_for_grid()

        def jitter(cas, xyzs):
            if draws is not None and draws.get("noise") is not None:
                return draws["noise"][cas].to(dev, torch.float32)
            return torch.rand_like(xyzs)
        if self.iter_density < 16:  # full refresh: every cell of every cascade
            cells = self._cells()
            c0, c1 = cells.shape[0] * rank // world, cells.shape[0] * (rank + 1) // world
            for cas in range(self.cascade):
                bound = min(2 ** cas, self.bound)
                half_grid_size = bound / H
                xyzs = cells[c0:c1] * (bound - half_grid_size)
                xyzs = xyzs + (jitter(cas, xyzs) * 2 - 1) * half_grid_size
                dens = sigma_of(xyzs).reshape(-1).detach().float() * self.density_scale
                tmp_grid[cas] = dens if world == 1 else gather(dens)
        else:  # partial refresh: H^3/4 uniform cells + H^3/4 currently occupied cells per cascade
            N = H ** 3 // 4 // world
            for cas in range(self.cascade):
                if draws is not None and draws.get("coords") is not None:
                    coords = draws["coords"][cas].to(dev)
                else:
                    coords = torch.randint(0, H, (N, 3), device=dev)
                indices = raymarching.morton3D(coords).long()
                # N occupied cells drawn uniformly with replacement (renderer.py:501-507: nonzero -> randint -> index), without
                # the read-back of the occupied count: the k-th occupied cell is where the running count reaches k + 1.
                # No occupied cell at all (the reference then adds no picks): the picks go to a spare slot behind the grid.
                running = torch.cumsum(self.density_grid[cas] > 0, 0)
                n_occ = running[-1]
                if draws is not None and draws.get("occ_k") is not None:
                    k = draws["occ_k"][cas].to(dev).long()
                else:
                    k = (torch.rand(N, device=dev, dtype=torch.float64) * n_occ).long().clamp_(max=(n_occ - 1).clamp(min=0))
                occ_indices = torch.searchsorted(running, k + 1)
                occ_indices = torch.where(n_occ > 0, occ_indices, torch.full_like(occ_indices, H ** 3))
                occ_coords = raymarching.morton3D_invert(occ_indices.clamp(max=H ** 3 - 1))
                indices = torch.cat([indices, occ_indices], dim=0)
                coords = torch.cat([coords, occ_coords], dim=0)
                xyzs = self._cell_centres(coords)
                bound = min(2 ** cas, self.bound)
                half_grid_size = bound / H
                xyzs = xyzs * (bound - half_grid_size)
                xyzs = xyzs + (jitter(cas, xyzs) * 2 - 1) * half_grid_size
                dens = sigma_of(xyzs).reshape(-1).detach().float() * self.density_scale
                tmp_c = -torch.ones(H ** 3 + 1, dtype=tmp_grid.dtype, device=dev)   # + the spare slot of the picks above
                # a cell drawn twice keeps the LARGER of its candidates: the reference's index assignment
                # (renderer.py:513) keeps an arbitrary one -- a race, so two runs of one training part ways at the
                # first partial refresh, and on several ranks each could keep a different one (the grids must stay
                # bit-identical across ranks).  The maximum is one of the outcomes the reference can produce, and it is
                # reproducible.
                if world == 1:
                    tmp_c.scatter_reduce_(0, indices, dens, reduce="amax", include_self=True)
                else:
                    tmp_c.scatter_reduce_(0, gather(indices), gather(dens), reduce="amax", include_self=True)
                tmp_grid[cas] = tmp_c[:H ** 3]
        # The same values as the reference's masked assignment / .item() / packbits / .item() sequence, with ONE host
        # read-back at the end instead of five (three of them hidden in the boolean-mask indexing): every kernel of the
        # refresh is enqueued before the host waits, the threshold min(density_thresh, mean) is taken on the device.
        valid_mask = (self.density_grid >= 0) & (tmp_grid >= 0)
        self.density_grid.copy_(torch.where(valid_mask, torch.maximum(self.density_grid * decay, tmp_grid),
                                            self.density_grid))
        mean_dev = torch.mean(self.density_grid.clamp(min=0)).reshape(1).float()
        self.iter_density += 1
        L.check(L.lib().tnl_packbits_dev(L.ptr(self.density_grid), L.u32(self.density_bitfield.numel()),
                                         L.f32(self.density_thresh), L.ptr(mean_dev), L.ptr(self.density_bitfield),
                                         L.stream()), "packbits_dev")
        self.density_bitfield[:0].zero_()    # the kernel wrote the bitfield behind torch's back: bump its version counter (an
        self._occ_box_key = None             # empty in-place op) so that whatever is cached against it is rebuilt (run_cuda's box)
        total_step = min(16, self.local_step)
        csum = self.step_counter[:total_step, 0].sum().reshape(1).double() if total_step > 0 else mean_dev.double() * 0
        mean_h, csum_h = torch.cat([mean_dev.double(), csum]).tolist()       # the refresh's one read-back (both exact in fp64)
        self.mean_density = mean_h
        if total_step > 0:
            self.mean_count = int(csum_h / total_step)
        self.local_step = 0

    def render(self, rays_o, rays_d, staged=False, max_ray_batch=4096, **kwargs):
        # reference: renderer.py:545-578
        _run = self.run_cuda if self.cuda_ray else self.run
        B, N = rays_o.shape[:2]
        device = rays_o.device
        if staged and not self.cuda_ray:
            depth = torch.empty((B, N), device=device)
            image = torch.empty((B, N, 3), device=device)
            for b in range(B):
                for head in range(0, N, max_ray_batch):
                    tail = min(head + max_ray_batch, N)
                    res = _run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail], **kwargs)
                    depth[b:b + 1, head:tail] = res['depth']
                    image[b:b + 1, head:tail] = res['image']
            return {'depth': depth, 'image': image}
        return _run(rays_o, rays_d, **kwargs)
