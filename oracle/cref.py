"""ctypes front-end of oracle/trinerflet_oracle.c (numpy in, numpy out).

TEST INFRASTRUCTURE ONLY -- see the header of trinerflet_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libtrinerflet_oracle.so")

WAVELETS = ("haar", "bior2.2", "bior4.4", "bior2.6", "bior6.8")


def build(force=False):
    """Compile the C oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "trinerflet_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def wavelet_taps(name):
    L = C.c_int()
    lo = np.zeros(18)
    hi = np.zeros(18)
    wid = lib().orc_wavelet_lookup(name.encode(), C.byref(L), _p(lo, C.c_double), _p(hi, C.c_double))
    if wid < 0:
        raise KeyError(name)
    return wid, L.value, lo[:L.value].copy(), hi[:L.value].copy()


def idwt_level(x, yh, wave, f64=False, taps_f32=True, ll_scale=2.0):
    """x:[S,n,n], yh:[S,3,n,n] -> [S,2n,2n] (triplane_encoder.py:379,392-394)."""
    wid = wavelet_taps(wave)[0]
    S, n = x.shape[0], x.shape[-1]
    if f64:
        x = np.ascontiguousarray(x, np.float64)
        yh = np.ascontiguousarray(yh, np.float64)
        out = np.empty((S, 2 * n, 2 * n), np.float64)
        lib().orc_idwt_level_f64(_p(x, C.c_double), _p(yh, C.c_double), S, n, wid, int(taps_f32),
                                 C.c_double(ll_scale), _p(out, C.c_double))
        return out
    x, yh = _f32(x), _f32(yh)
    out = np.empty((S, 2 * n, 2 * n), np.float32)
    lib().orc_idwt_level_f32(_p(x, C.c_float), _p(yh, C.c_float), S, n, wid, _p(out, C.c_float))
    return out


def idwt_level_adj(dout, wave, f64=False, taps_f32=True, ll_scale=2.0):
    """dout:[S,2n,2n] -> (dx:[S,n,n], dyh:[S,3,n,n])."""
    wid = wavelet_taps(wave)[0]
    S, n = dout.shape[0], dout.shape[-1] // 2
    if f64:
        dout = np.ascontiguousarray(dout, np.float64)
        dx = np.empty((S, n, n), np.float64)
        dyh = np.empty((S, 3, n, n), np.float64)
        lib().orc_idwt_level_adj_f64(_p(dout, C.c_double), S, n, wid, int(taps_f32),
                                     C.c_double(ll_scale), _p(dx, C.c_double), _p(dyh, C.c_double))
        return dx, dyh
    dout = _f32(dout)
    dx = np.empty((S, n, n), np.float32)
    dyh = np.empty((S, 3, n, n), np.float32)
    lib().orc_idwt_level_adj_f32(_p(dout, C.c_float), S, n, wid, _p(dx, C.c_float), _p(dyh, C.c_float))
    return dx, dyh


def build_planes(ll, coefs, wave, f64=False):
    """ll:[3,C,n0,n0], coefs: list of [3,C,3,n,n] coarse->fine -> planes [3,C,R,R]
    (TriPlaneVolume.build_planes, triplane_encoder.py:364-405)."""
    P, Cc = ll.shape[:2]
    x = np.asarray(ll).reshape(P * Cc, ll.shape[-2], ll.shape[-1])
    for yh in coefs:
        n = x.shape[-1]
        x = idwt_level(x, np.asarray(yh).reshape(P * Cc, 3, n, n), wave, f64=f64)
    return x.reshape(P, Cc, x.shape[-1], x.shape[-1])


def build_planes_adj(dplanes, levels, wave, f64=False):
    """VJP of build_planes: dplanes [3,C,R,R] -> (dll, [dcoef_0..dcoef_{J-1}])."""
    P, Cc, R = dplanes.shape[:3]
    g = np.asarray(dplanes).reshape(P * Cc, R, R)
    dcoefs = []
    for _ in range(levels):
        g, dyh = idwt_level_adj(g, wave, f64=f64)
        n = g.shape[-1]
        dcoefs.append(dyh.reshape(P, Cc, 3, n, n))
    return g.reshape(P, Cc, g.shape[-1], g.shape[-1]), dcoefs[::-1]


def triplane_sample(planes, xyz, bound):
    planes, xyz = _f32(planes), _f32(xyz)
    _, Cc, R, _ = planes.shape
    N = xyz.shape[0]
    out = np.empty((N, 3 * Cc), np.float32)
    lib().orc_triplane_sample(_p(planes, C.c_float), _p(xyz, C.c_float), C.c_float(bound), N, Cc, R,
                              _p(out, C.c_float))
    return out


def triplane_sample_bwd(dout, xyz, bound, Cc, R):
    dout, xyz = _f32(dout), _f32(xyz)
    N = xyz.shape[0]
    dpl = np.zeros((3, Cc, R, R), np.float64)
    lib().orc_triplane_sample_bwd(_p(dout, C.c_float), _p(xyz, C.c_float), C.c_float(bound), N, Cc, R,
                                  _p(dpl, C.c_double))
    return dpl


def sh4(dirs):
    dirs = _f32(dirs)
    out = np.empty((dirs.shape[0], 16), np.float32)
    lib().orc_sh4(_p(dirs, C.c_float), dirs.shape[0], _p(out, C.c_float))
    return out


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    rays_o, rays_d, aabb = _f32(rays_o), _f32(rays_d), _f32(aabb)
    N = rays_o.shape[0]
    nears = np.empty(N, np.float32)
    fars = np.empty(N, np.float32)
    lib().orc_near_far_from_aabb(_p(rays_o, C.c_float), _p(rays_d, C.c_float), _p(aabb, C.c_float),
                                 C.c_uint32(N), C.c_float(min_near), _p(nears, C.c_float), _p(fars, C.c_float))
    return nears, fars


def morton3D(coords):
    coords = np.ascontiguousarray(coords, np.int32)
    out = np.empty(coords.shape[0], np.int32)
    lib().orc_morton3D(_p(coords, C.c_int), C.c_uint32(coords.shape[0]), _p(out, C.c_int))
    return out


def morton3D_invert(indices):
    indices = np.ascontiguousarray(indices, np.int32)
    out = np.empty((indices.shape[0], 3), np.int32)
    lib().orc_morton3D_invert(_p(indices, C.c_int), C.c_uint32(indices.shape[0]), _p(out, C.c_int))
    return out


def packbits(grid, thresh):
    grid = _f32(grid)
    N = grid.size // 8
    out = np.empty(N, np.uint8)
    lib().orc_packbits(_p(grid, C.c_float), C.c_uint32(N), C.c_float(thresh), _p(out, C.c_uint8))
    return out


def march_rays_train(rays_o, rays_d, bound, bitfield, Cas, H, nears, fars, noises, M,
                     dt_gamma=0.0, max_steps=1024, counter=None):
    """Returns xyzs[M,3], dirs[M,3], deltas[M,2], rays[N,3], counter[2] (ray-id order packing)."""
    rays_o, rays_d = _f32(rays_o), _f32(rays_d)
    nears, fars, noises = _f32(nears), _f32(fars), _f32(noises)
    bitfield = np.ascontiguousarray(bitfield, np.uint8)
    N = rays_o.shape[0]
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    rays = np.empty((N, 3), np.int32)
    if counter is None:
        counter = np.zeros(2, np.int32)
    lib().orc_march_rays_train(_p(rays_o, C.c_float), _p(rays_d, C.c_float), _p(bitfield, C.c_uint8),
                               C.c_float(bound), C.c_float(dt_gamma), C.c_uint32(max_steps), C.c_uint32(N),
                               C.c_uint32(Cas), C.c_uint32(H), C.c_uint32(M), _p(nears, C.c_float),
                               _p(fars, C.c_float), _p(xyzs, C.c_float), _p(dirs, C.c_float),
                               _p(deltas, C.c_float), _p(rays, C.c_int), _p(counter, C.c_int),
                               _p(noises, C.c_float))
    return xyzs, dirs, deltas, rays, counter


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, T_thresh=1e-4):
    sigmas, rgbs, deltas = _f32(sigmas), _f32(rgbs), _f32(deltas)
    rays = np.ascontiguousarray(rays, np.int32)
    M, N = sigmas.shape[0], rays.shape[0]
    ws = np.empty(N, np.float32)
    depth = np.empty(N, np.float32)
    image = np.empty((N, 3), np.float32)
    lib().orc_composite_rays_train_forward(_p(sigmas, C.c_float), _p(rgbs, C.c_float), _p(deltas, C.c_float),
                                           _p(rays, C.c_int), C.c_uint32(M), C.c_uint32(N), C.c_float(T_thresh),
                                           _p(ws, C.c_float), _p(depth, C.c_float), _p(image, C.c_float))
    return ws, depth, image


def composite_rays_train_backward(grad_ws, grad_image, sigmas, rgbs, deltas, rays, ws, image, T_thresh=1e-4):
    grad_ws, grad_image = _f32(grad_ws), _f32(grad_image)
    sigmas, rgbs, deltas, ws, image = _f32(sigmas), _f32(rgbs), _f32(deltas), _f32(ws), _f32(image)
    rays = np.ascontiguousarray(rays, np.int32)
    M, N = sigmas.shape[0], rays.shape[0]
    gs = np.zeros(M, np.float32)
    gc = np.zeros((M, 3), np.float32)
    lib().orc_composite_rays_train_backward(_p(grad_ws, C.c_float), _p(grad_image, C.c_float),
                                            _p(sigmas, C.c_float), _p(rgbs, C.c_float), _p(deltas, C.c_float),
                                            _p(rays, C.c_int), _p(ws, C.c_float), _p(image, C.c_float),
                                            C.c_uint32(M), C.c_uint32(N), C.c_float(T_thresh),
                                            _p(gs, C.c_float), _p(gc, C.c_float))
    return gs, gc


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, bitfield, Cas, H, nears, fars,
               noises, align=-1, dt_gamma=0.0, max_steps=1024):
    rays_o, rays_d = _f32(rays_o), _f32(rays_d)
    nears, fars, noises, rays_t = _f32(nears), _f32(fars), _f32(noises), _f32(rays_t)
    rays_alive = np.ascontiguousarray(rays_alive, np.int32)
    bitfield = np.ascontiguousarray(bitfield, np.uint8)
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    lib().orc_march_rays(C.c_uint32(n_alive), C.c_uint32(n_step), _p(rays_alive, C.c_int), _p(rays_t, C.c_float),
                         _p(rays_o, C.c_float), _p(rays_d, C.c_float), C.c_float(bound), C.c_float(dt_gamma),
                         C.c_uint32(max_steps), C.c_uint32(Cas), C.c_uint32(H), _p(bitfield, C.c_uint8),
                         _p(nears, C.c_float), _p(fars, C.c_float), _p(xyzs, C.c_float), _p(dirs, C.c_float),
                         _p(deltas, C.c_float), _p(noises, C.c_float))
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image,
                   T_thresh=1e-2):
    """In place on rays_alive, rays_t, weights_sum, depth, image (numpy arrays of the right dtype)."""
    sigmas, rgbs, deltas = _f32(sigmas), _f32(rgbs), _f32(deltas)
    for a, t in ((rays_alive, np.int32), (rays_t, np.float32), (weights_sum, np.float32),
                 (depth, np.float32), (image, np.float32)):
        assert a.dtype == t and a.flags.c_contiguous
    lib().orc_composite_rays(C.c_uint32(n_alive), C.c_uint32(n_step), C.c_float(T_thresh), _p(rays_alive, C.c_int),
                             _p(rays_t, C.c_float), _p(sigmas, C.c_float), _p(rgbs, C.c_float),
                             _p(deltas, C.c_float), _p(weights_sum, C.c_float), _p(depth, C.c_float),
                             _p(image, C.c_float))


def get_rays(poses, intrinsics, H, W, pix):
    """utils.py:65-149 for flat pixel ids pix[n] = b*H*W + y*W + x -> rays_o, rays_d [N,3]."""
    poses = _f32(poses).reshape(-1, 16)
    intrinsics = _f32(intrinsics)
    pix = np.ascontiguousarray(pix, np.int64)
    N = pix.shape[0]
    rays_o = np.empty((N, 3), np.float32)
    rays_d = np.empty((N, 3), np.float32)
    lib().orc_get_rays(_p(poses, C.c_float), _p(intrinsics, C.c_float), C.c_uint32(H), C.c_uint32(W),
                       _p(pix, C.c_int64), C.c_uint64(N), _p(rays_o, C.c_float), _p(rays_d, C.c_float))
    return rays_o, rays_d


def permute_index(g, total, key):
    f = lib().orc_permute_index
    f.restype = C.c_uint64
    return np.array([f(C.c_uint64(int(v)), C.c_uint64(int(total)), C.c_uint64(int(key))) for v in np.atleast_1d(g)],
                    np.int64)
