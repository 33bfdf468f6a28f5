"""`_raymarching` -- the native module name the reference binds (aux_libs/raymarching/raymarching.py:9-12,
`import _raymarching as _backend`; pybind definitions aux_libs/raymarching/src/bindings.cpp:5-18), served by
libtrinerflet_hip.so.

The ten functions have the names, argument order and conventions of aux_libs/raymarching/src/raymarching.h:7-17:
all return None, tensors are passed by the caller (which allocates every output, SURVEY.md 8(b) "Ownership"),
scalars are uint32 / float.  With this directory on sys.path (trinerflet_amd.install_dropin() puts it there) the
reference's own raymarching.py -- its nine autograd Functions unchanged -- runs on the MI355X kernels.

Differences from the CUDA module, all inherent to the C ABI underneath (include/trinerflet_hip.h):
  * fp32 / int32 / uint8 tensors only (the reference's wrappers force fp32 with custom_fwd(cast_inputs=float32));
    other dtypes raise TypeError instead of being dispatched;
  * kernels launch on torch's current stream, not the legacy default stream;
  * march_rays_train packs the rays in ray-id order (one of the reference's possible atomic arrival orders) and
    allocates its own int32 scratch buffer.
"""
import torch

from trinerflet_amd import _lib as L

__all__ = ["near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "march_rays_train",
           "composite_rays_train_forward", "composite_rays_train_backward", "march_rays", "composite_rays"]


def _chk(dtype, *tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("_raymarching: tensor is not on a HIP device (no CPU fallback exists)")
        if t.dtype != dtype:
            raise TypeError(f"_raymarching: expected {dtype}, got {t.dtype}")
        if not t.is_contiguous():
            raise RuntimeError("_raymarching: tensor must be contiguous")


def _f(*t):
    _chk(torch.float32, *t)


def _i(*t):
    _chk(torch.int32, *t)


def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
    """raymarching.h:7"""
    _f(rays_o, rays_d, aabb, nears, fars)
    L.check(L.lib().tnl_near_far_from_aabb(L.ptr(rays_o), L.ptr(rays_d), L.ptr(aabb), L.u32(N), L.f32(min_near),
                                           L.ptr(nears), L.ptr(fars), L.stream()), "near_far_from_aabb")


def sph_from_ray(rays_o, rays_d, radius, N, coords):
    """raymarching.h:8"""
    _f(rays_o, rays_d, coords)
    L.check(L.lib().tnl_sph_from_ray(L.ptr(rays_o), L.ptr(rays_d), L.f32(radius), L.u32(N), L.ptr(coords),
                                     L.stream()), "sph_from_ray")


def morton3D(coords, N, indices):
    """raymarching.h:9"""
    _i(coords, indices)
    L.check(L.lib().tnl_morton3D(L.ptr(coords), L.u32(N), L.ptr(indices), L.stream()), "morton3D")


def morton3D_invert(indices, N, coords):
    """raymarching.h:10"""
    _i(indices, coords)
    L.check(L.lib().tnl_morton3D_invert(L.ptr(indices), L.u32(N), L.ptr(coords), L.stream()), "morton3D_invert")


def packbits(grid, N, density_thresh, bitfield):
    """raymarching.h:11"""
    _f(grid)
    _chk(torch.uint8, bitfield)
    L.check(L.lib().tnl_packbits(L.ptr(grid), L.u32(N), L.f32(density_thresh), L.ptr(bitfield), L.stream()),
            "packbits")


def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas,
                     rays, counter, noises):
    """raymarching.h:13"""
    _f(rays_o, rays_d, nears, fars, xyzs, dirs, deltas, noises)
    _i(rays, counter)
    _chk(torch.uint8, grid)
    if M >= 2 ** 32:
        raise ValueError(f"march_rays_train: M = {M} does not fit the uint32 of raymarching.h:13")
    lib = L.lib()
    nws = lib.tnl_march_rays_train_workspace_rec(L.u32(N), L.u32(max_steps)) or \
        lib.tnl_march_rays_train_workspace(L.u32(N))
    ws = torch.empty(nws, dtype=torch.int32, device=rays_o.device)
    L.check(lib.tnl_march_rays_train(L.ptr(rays_o), L.ptr(rays_d), L.ptr(grid), L.f32(bound), L.f32(dt_gamma),
                                     L.u32(max_steps), L.u32(N), L.u32(C), L.u32(H), L.u32(M), L.ptr(nears),
                                     L.ptr(fars), L.ptr(xyzs), L.ptr(dirs), L.ptr(deltas), L.ptr(rays),
                                     L.ptr(counter), L.ptr(noises), L.ptr(ws), L.u32(nws), L.stream()),
            "march_rays_train")


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image):
    """raymarching.h:14"""
    _f(sigmas, rgbs, deltas, weights_sum, depth, image)
    _i(rays)
    L.check(L.lib().tnl_composite_rays_train_forward(L.ptr(sigmas), L.ptr(rgbs), L.ptr(deltas), L.ptr(rays), L.u32(M),
                                                     L.u32(N), L.f32(T_thresh), L.ptr(weights_sum), L.ptr(depth),
                                                     L.ptr(image), L.stream()), "composite_rays_train_forward")


def composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N,
                                  T_thresh, grad_sigmas, grad_rgbs):
    """raymarching.h:15"""
    _f(grad_weights_sum, grad_image, sigmas, rgbs, deltas, weights_sum, image, grad_sigmas, grad_rgbs)
    _i(rays)
    L.check(L.lib().tnl_composite_rays_train_backward(L.ptr(grad_weights_sum), L.ptr(grad_image), L.ptr(sigmas),
                                                      L.ptr(rgbs), L.ptr(deltas), L.ptr(rays), L.ptr(weights_sum),
                                                      L.ptr(image), L.u32(M), L.u32(N), L.f32(T_thresh),
                                                      L.ptr(grad_sigmas), L.ptr(grad_rgbs), L.stream()),
            "composite_rays_train_backward")


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears,
               fars, xyzs, dirs, deltas, noises):
    """raymarching.h:16"""
    _f(rays_t, rays_o, rays_d, nears, fars, xyzs, dirs, deltas, noises)
    _i(rays_alive)
    _chk(torch.uint8, grid)
    L.check(L.lib().tnl_march_rays(L.u32(n_alive), L.u32(n_step), L.ptr(rays_alive), L.ptr(rays_t), L.ptr(rays_o),
                                   L.ptr(rays_d), L.f32(bound), L.f32(dt_gamma), L.u32(max_steps), L.u32(C), L.u32(H),
                                   L.ptr(grid), L.ptr(nears), L.ptr(fars), L.ptr(xyzs), L.ptr(dirs), L.ptr(deltas),
                                   L.ptr(noises), L.stream()), "march_rays")


def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    """raymarching.h:17"""
    _f(rays_t, sigmas, rgbs, deltas, weights_sum, depth, image)
    _i(rays_alive)
    L.check(L.lib().tnl_composite_rays(L.u32(n_alive), L.u32(n_step), L.f32(T_thresh), L.ptr(rays_alive),
                                       L.ptr(rays_t), L.ptr(sigmas), L.ptr(rgbs), L.ptr(deltas), L.ptr(weights_sum),
                                       L.ptr(depth), L.ptr(image), L.stream()), "composite_rays")
