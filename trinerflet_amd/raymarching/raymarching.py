"""Mirror of aux_libs/raymarching/raymarching.py (reference) on top of libtrinerflet_hip.so.

Same function names, argument order, defaults and return values as the reference's nine
autograd Functions (raymarching.py:19-373).  Inputs are moved to the current HIP device and
made contiguous fp32 like the reference wrappers do (custom_fwd(cast_inputs=float32), .cuda(),
.contiguous()).  Kernels launch on torch's current stream.
"""
import torch
from torch.autograd import Function

from .. import _lib as L

__all__ = ["near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "occupied_box", "clip_fars",
           "march_rays_train", "count_form", "side_caps", "composite_rays_train", "march_rays", "composite_rays", "compact_rays"]


def _f32c(t):
    if not t.is_cuda:
        t = t.cuda()
    return t.detach().to(torch.float32).contiguous()


class _near_far_from_aabb(Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        # reference: raymarching.py:19-47
        rays_o = _f32c(rays_o).view(-1, 3)
        rays_d = _f32c(rays_d).view(-1, 3)
        aabb = _f32c(aabb)
        N = rays_o.shape[0]
        nears = torch.empty(N, dtype=torch.float32, device=rays_o.device)
        fars = torch.empty(N, dtype=torch.float32, device=rays_o.device)
        L.check(L.lib().tnl_near_far_from_aabb(L.ptr(rays_o), L.ptr(rays_d), L.ptr(aabb), L.u32(N),
                                               L.f32(min_near), L.ptr(nears), L.ptr(fars), L.stream()),
                "near_far_from_aabb")
        return nears, fars


near_far_from_aabb = _near_far_from_aabb.apply


class _sph_from_ray(Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, radius):
        # reference: raymarching.py:52-78
        rays_o = _f32c(rays_o).view(-1, 3)
        rays_d = _f32c(rays_d).view(-1, 3)
        N = rays_o.shape[0]
        coords = torch.empty(N, 2, dtype=torch.float32, device=rays_o.device)
        L.check(L.lib().tnl_sph_from_ray(L.ptr(rays_o), L.ptr(rays_d), L.f32(radius), L.u32(N), L.ptr(coords),
                                         L.stream()), "sph_from_ray")
        return coords


sph_from_ray = _sph_from_ray.apply


class _morton3D(Function):
    @staticmethod
    def forward(ctx, coords):
        # reference: raymarching.py:83-102
        if not coords.is_cuda:
            coords = coords.cuda()
        coords = coords.int().contiguous()
        N = coords.shape[0]
        indices = torch.empty(N, dtype=torch.int32, device=coords.device)
        L.check(L.lib().tnl_morton3D(L.ptr(coords), L.u32(N), L.ptr(indices), L.stream()), "morton3D")
        return indices


morton3D = _morton3D.apply


class _morton3D_invert(Function):
    @staticmethod
    def forward(ctx, indices):
        # reference: raymarching.py:106-124
        if not indices.is_cuda:
            indices = indices.cuda()
        indices = indices.int().contiguous()
        N = indices.shape[0]
        coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
        L.check(L.lib().tnl_morton3D_invert(L.ptr(indices), L.u32(N), L.ptr(coords), L.stream()), "morton3D_invert")
        return coords


morton3D_invert = _morton3D_invert.apply


class _packbits(Function):
    @staticmethod
    def forward(ctx, grid, thresh, bitfield=None):
        # reference: raymarching.py:129-153
        grid = _f32c(grid)
        C, H3 = grid.shape[0], grid.shape[1]
        N = C * H3 // 8
        if bitfield is None:
            bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
        L.check(L.lib().tnl_packbits(L.ptr(grid), L.u32(N), L.f32(thresh), L.ptr(bitfield), L.stream()), "packbits")
        bitfield[:0].zero_()      # the kernel wrote it in place: bump torch's version counter (an empty in-place op), so that
        return bitfield           # anything cached against the bitfield's version (run_cuda's occupied box) notices


packbits = _packbits.apply


class _march_rays_train(Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024, noises=None,
                zero_fill=True, sort=None):
        # reference: raymarching.py:161-233.  Extra trailing `noises` ([N] in [0,1)) lets a caller supply
        # the perturbation explicitly (used by tests/bench for seeded parity); None = reference behaviour.
        # sort=(R, workspace): the writing kernel also counts the samples per plane tile for the plane-gradient sort
        # (nerf/field.py plane_grad_sort_counted finishes it).
        # zero_fill=False skips the reference's zero fill of the padded sample buffers (:205-207) for a caller whose
        # consumers never read rows past counter[0] (TrainStep: every kernel takes the count): 32 B x M less to write.
        rays_o = _f32c(rays_o).view(-1, 3)
        rays_d = _f32c(rays_d).view(-1, 3)
        nears, fars = _f32c(nears), _f32c(fars)
        if not density_bitfield.is_cuda:
            density_bitfield = density_bitfield.cuda()
        density_bitfield = density_bitfield.contiguous()
        dev = rays_o.device
        N = rays_o.shape[0]
        M = N * max_steps
        if not force_all_rays and mean_count > 0:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count
        if M >= 2 ** 32 or N >= 2 ** 31:
            # the C ABI takes the sample budget as uint32 (raymarching.h:13 does too); do not truncate silently
            raise ValueError(f"march_rays_train: sample budget M = {M} (N = {N}, max_steps = {max_steps}) does not fit "
                             "the 32-bit sample index of the kernels; march fewer rays per call")
        alloc = torch.zeros if zero_fill else torch.empty
        xyzs = alloc(M, 3, dtype=torch.float32, device=dev)
        dirs = alloc(M, 3, dtype=torch.float32, device=dev)
        deltas = alloc(M, 2, dtype=torch.float32, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
        if noises is not None:
            noises = _f32c(noises)
        elif perturb:
            noises = torch.rand(N, dtype=torch.float32, device=dev)
        else:
            noises = torch.zeros(N, dtype=torch.float32, device=dev)
        lib = L.lib()
        # scratch incl. the per-sample t record (N * max_steps floats, never filled beyond the samples that exist): the
        # rays are marched once; falls back to the two-march form if that size does not fit 32 bits
        nws = lib.tnl_march_rays_train_workspace_rec(L.u32(N), L.u32(max_steps)) or \
            lib.tnl_march_rays_train_workspace(L.u32(N))
        ws = torch.empty(nws, dtype=torch.int32, device=dev)
        args = (L.ptr(rays_o), L.ptr(rays_d), L.ptr(density_bitfield), L.f32(bound), L.f32(dt_gamma), L.u32(max_steps),
                L.u32(N), L.u32(C), L.u32(H), L.u32(M), L.ptr(nears), L.ptr(fars), L.ptr(xyzs), L.ptr(dirs),
                L.ptr(deltas), L.ptr(rays), L.ptr(step_counter), L.ptr(noises), L.ptr(ws), L.u32(nws))
        if sort is None:
            L.check(lib.tnl_march_rays_train(*args, L.stream()), "march_rays_train")
        else:
            L.check(lib.tnl_march_rays_train_binned(*args, L.u32(sort[0]), L.ptr(sort[1]), L.stream()),
                    "march_rays_train_binned")
        if force_all_rays or mean_count <= 0:
            m = step_counter[0].item()  # D2H copy, as in the reference (:224)
            if align > 0:
                m += align - m % align
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
        return xyzs, dirs, deltas, rays


march_rays_train = _march_rays_train.apply


class count_form:
    """with raymarching.count_form(1): ... -- the count pass of march_rays_train inside the block walks one ray per lane
    (tnl_march_count_form: the small-footprint form for a march enqueued beside other kernels); 0 = the default, one
    wavefront per ray.  A switch of the library for the calling host thread, restored on exit."""

    def __init__(self, form):
        self.form = int(form)

    def __enter__(self):
        self.prev = L.lib().tnl_march_count_form(L.i32(self.form))
        return self

    def __exit__(self, *exc):
        L.lib().tnl_march_count_form(L.i32(self.prev))
        return False


class side_caps:
    """with raymarching.side_caps(emit_blocks, fill_blocks): ... -- the emit pass of march_rays_train and the fill pass of
    the plane-gradient tile sort launched inside the block keep to that many workgroups (tnl_march_emit_cap,
    tnl_plane_grad_fill_cap; 0 = full width): for work enqueued beside other kernels.  Per host thread, restored on exit."""

    def __init__(self, emit_blocks, fill_blocks):
        self.caps = (int(emit_blocks), int(fill_blocks))

    def __enter__(self):
        self.prev = (L.lib().tnl_march_emit_cap(L.i32(self.caps[0])), L.lib().tnl_plane_grad_fill_cap(L.i32(self.caps[1])))
        return self

    def __exit__(self, *exc):
        L.lib().tnl_march_emit_cap(L.i32(self.prev[0]))
        L.lib().tnl_plane_grad_fill_cap(L.i32(self.prev[1]))
        return False


class _composite_rays_train(Function):
    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
        # reference: raymarching.py:238-268
        sigmas, rgbs, deltas = _f32c(sigmas), _f32c(rgbs), _f32c(deltas)
        rays = rays.contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        dev = sigmas.device
        weights_sum = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        L.check(L.lib().tnl_composite_rays_train_forward(L.ptr(sigmas), L.ptr(rgbs), L.ptr(deltas), L.ptr(rays),
                                                         L.u32(M), L.u32(N), L.f32(T_thresh), L.ptr(weights_sum),
                                                         L.ptr(depth), L.ptr(image), L.stream()),
                "composite_rays_train_forward")
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        ctx.dims = [M, N, T_thresh]
        return weights_sum, depth, image

    @staticmethod
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        # reference: raymarching.py:270-288 (grad_depth is not propagated there either)
        sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N, T_thresh = ctx.dims
        grad_weights_sum = grad_weights_sum.to(torch.float32).contiguous()
        grad_image = grad_image.to(torch.float32).contiguous()
        grad_sigmas = torch.zeros_like(sigmas)
        grad_rgbs = torch.zeros_like(rgbs)
        L.check(L.lib().tnl_composite_rays_train_backward(L.ptr(grad_weights_sum), L.ptr(grad_image), L.ptr(sigmas),
                                                          L.ptr(rgbs), L.ptr(deltas), L.ptr(rays),
                                                          L.ptr(weights_sum), L.ptr(image), L.u32(M), L.u32(N),
                                                          L.f32(T_thresh), L.ptr(grad_sigmas), L.ptr(grad_rgbs),
                                                          L.stream()), "composite_rays_train_backward")
        return grad_sigmas, grad_rgbs, None, None, None


composite_rays_train = _composite_rays_train.apply


class _march_rays(Function):
    @staticmethod
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far,
                align=-1, perturb=False, dt_gamma=0, max_steps=1024):
        # reference: raymarching.py:297-346
        rays_o = _f32c(rays_o).view(-1, 3)
        rays_d = _f32c(rays_d).view(-1, 3)
        dev = rays_o.device
        M = n_alive * n_step
        if align > 0:
            M += align - (M % align)
        xyzs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
        dirs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
        deltas = torch.zeros(M, 2, dtype=torch.float32, device=dev)
        if perturb:
            noises = torch.rand(n_alive, dtype=torch.float32, device=dev)
        else:
            noises = torch.zeros(n_alive, dtype=torch.float32, device=dev)
        L.check(L.lib().tnl_march_rays(L.u32(n_alive), L.u32(n_step), L.ptr(rays_alive), L.ptr(rays_t), L.ptr(rays_o),
                                       L.ptr(rays_d), L.f32(bound), L.f32(dt_gamma), L.u32(max_steps), L.u32(C),
                                       L.u32(H), L.ptr(density_bitfield), L.ptr(near), L.ptr(far), L.ptr(xyzs),
                                       L.ptr(dirs), L.ptr(deltas), L.ptr(noises), L.stream()), "march_rays")
        return xyzs, dirs, deltas


march_rays = _march_rays.apply


class _composite_rays(Function):
    @staticmethod
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image,
                T_thresh=1e-2):
        # reference: raymarching.py:353-370 (in place; returns an empty tuple)
        sigmas, rgbs = _f32c(sigmas), _f32c(rgbs)
        L.check(L.lib().tnl_composite_rays(L.u32(n_alive), L.u32(n_step), L.f32(T_thresh), L.ptr(rays_alive),
                                           L.ptr(rays_t), L.ptr(sigmas), L.ptr(rgbs), L.ptr(deltas),
                                           L.ptr(weights_sum), L.ptr(depth), L.ptr(image), L.stream()),
                "composite_rays")
        return tuple()


composite_rays = _composite_rays.apply


def compact_rays(rays_alive, n_alive=None):
    """Device-side `rays_alive[rays_alive >= 0]` (renderer.py:364) without a boolean-mask kernel chain.
    Returns (compacted int32 tensor of the same capacity, device int32 scalar with the survivor count)."""
    if n_alive is None:
        n_alive = rays_alive.shape[0]
    out = torch.empty_like(rays_alive)
    n_out = torch.empty(1, dtype=torch.int32, device=rays_alive.device)
    ws = torch.empty((n_alive + 255) // 256 + 1, dtype=torch.int32, device=rays_alive.device)
    L.check(L.lib().tnl_compact_rays(L.ptr(rays_alive), L.u32(n_alive), L.ptr(out), L.ptr(n_out), L.ptr(ws),
                                     L.stream()), "compact_rays")
    return out, n_out


def occupied_box(density_bitfield, cascade, H, bound):
    """Device tensor [6]: world-space box of the occupied cells of `density_bitfield` grown by one cell (csrc/raymarch.hip
    k_occupied_box); faces at the volume boundary are infinite.  Valid for this bitfield only.  No host read-back."""
    bits = density_bitfield.contiguous().view(cascade, -1)
    dev = bits.device
    scratch = torch.empty(cascade * 6, dtype=torch.int32, device=dev)
    box = torch.empty(6, dtype=torch.float32, device=dev)
    L.check(L.lib().tnl_occupied_box(L.ptr(bits), L.u32(bits.shape[1]), L.u32(cascade), L.u32(H), L.f32(bound),
                                     L.ptr(scratch), L.ptr(box), L.stream()), "occupied_box")
    return box


def clip_fars(rays_o, rays_d, fars, box):
    """fars limited to each ray's exit from `box` (occupied_box): marching up to that value finds the same samples, to
    the bit, as marching up to `fars` -- no occupied cell lies behind it -- without probing the empty cells there."""
    rays_o, rays_d, fars = _f32c(rays_o).view(-1, 3), _f32c(rays_d).view(-1, 3), _f32c(fars)
    out = torch.empty_like(fars)
    L.check(L.lib().tnl_clip_fars(L.ptr(rays_o), L.ptr(rays_d), L.ptr(fars), L.ptr(box), L.u32(fars.numel()), L.ptr(out),
                                  L.stream()), "clip_fars")
    return out
