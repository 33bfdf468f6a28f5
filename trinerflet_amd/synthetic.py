"""Seeded synthetic inputs shared by bench.py and the parity tests.

numpy only: usable by the oracle-side and the GPU-side of a test alike.
"""
import numpy as np


def sphere_bitfield(H=128, cascades=2, bound=1.5, r_out=0.8, r_in=0.0):
    """Analytic occupancy: cell occupied iff its centre radius is in [r_in, r_out].
    Layout = density_bitfield of the reference: bit (cas*H^3 + morton3D(x,y,z))."""
    idx = np.arange(H, dtype=np.uint32)

    def expand(v):
        v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
        v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
        v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
        v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
        return v

    ex = expand(idx)
    mort = (ex[:, None, None] | (ex[None, :, None] << 1) | (ex[None, None, :] << 2)).astype(np.int64)
    bits = np.zeros(cascades * H ** 3, np.uint8)
    for cas in range(cascades):
        b = min(2.0 ** cas, bound)
        c = ((idx.astype(np.float64) + 0.5) / H * 2 - 1) * b
        rr = np.sqrt(c[:, None, None] ** 2 + c[None, :, None] ** 2 + c[None, None, :] ** 2)
        occ = (rr <= r_out) & (rr >= r_in)
        bits[cas * H ** 3 + mort.reshape(-1)] = occ.reshape(-1)
    return np.packbits(bits.reshape(-1, 8), axis=1, bitorder="little").reshape(-1)


def hemisphere_poses(n, radius=4.0311, seed=0):
    """n camera-to-world matrices on the upper hemisphere looking at the origin, in the NGP frame
    (provider.py:23-31 nerf_matrix_to_ngp with scale=1, offset=0 is folded in: it permutes axes only,
    which a camera set drawn uniformly on the hemisphere absorbs)."""
    rng = np.random.default_rng(seed)
    poses = np.zeros((n, 4, 4), np.float32)
    for i in range(n):
        v = rng.standard_normal(3)
        v /= np.linalg.norm(v)
        v[1] = abs(v[1])  # y up in the NGP frame
        c = v * radius
        fwd = -v  # camera looks at the origin (+z forward, matches get_rays' zs = +1)
        up = np.array([0.0, 1.0, 0.0])
        right = np.cross(up, fwd)
        right /= np.linalg.norm(right) + 1e-12
        up2 = np.cross(fwd, right)
        poses[i, :3, 0], poses[i, :3, 1], poses[i, :3, 2], poses[i, :3, 3] = right, up2, fwd, c
        poses[i, 3, 3] = 1
    return poses


def get_rays(poses, pix, H=800, W=800, camera_angle_x=0.6911):
    """get_rays semantics (reconstruction/nerf/utils.py:65-149): pixel centre +0.5, normalised dirs,
    rays_d = dirs @ R^T, rays_o = t.  pix: [N,2] = (camera index, flat pixel index)."""
    fl = W / (2 * np.tan(camera_angle_x / 2))
    cam, p = pix[:, 0], pix[:, 1]
    i = (p % W).astype(np.float32) + 0.5
    j = (p // W).astype(np.float32) + 0.5
    d = np.stack([(i - W / 2) / fl, (j - H / 2) / fl, np.ones_like(i)], -1).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    R = poses[cam, :3, :3]
    rays_d = np.einsum("nij,nj->ni", R, d).astype(np.float32)
    rays_o = poses[cam, :3, 3].astype(np.float32)
    return rays_o, rays_d


def training_rays(N, n_cams=100, seed=0, H=800, W=800):
    poses = hemisphere_poses(n_cams, seed=seed)
    rng = np.random.default_rng(seed + 1)
    flat = np.unique(rng.integers(0, n_cams * H * W, size=N + N // 8 + 16))   # distinct pixels, O(N)
    flat = rng.permutation(flat)[:N]
    pix = np.stack([flat // (H * W), flat % (H * W)], -1)
    return get_rays(poses, pix, H, W)


def target_colors(rays_d):
    """Analytic ground-truth colours in [0,1] (a smooth function of the ray direction): gives the
    benchmark a non-trivial loss and gradients without a dataset."""
    d = np.asarray(rays_d, np.float32)
    return (0.5 + 0.5 * np.sin(3.0 * d + np.array([0.0, 1.0, 2.0], np.float32))).astype(np.float32)


def sphere_scene_rgba(rays_o, rays_d, radius=0.6):
    """Analytic scene for end-to-end runs without a dataset: an opaque sphere shaded by its normal
    (rgb = 0.5 + 0.5 n) with alpha = 1 where the ray hits it, 0 elsewhere -> [N,4] like a Blender RGBA pixel."""
    o, d = np.asarray(rays_o, np.float64), np.asarray(rays_d, np.float64)
    b = (o * d).sum(-1)
    disc = b * b - ((o * o).sum(-1) - radius ** 2)
    hit = disc > 0
    t = -b - np.sqrt(np.clip(disc, 0, None))
    n = (o + t[:, None] * d) / radius
    rgb = (0.5 + 0.5 * n) * hit[:, None]
    return np.concatenate([rgb, hit[:, None].astype(np.float64)], -1).astype(np.float32)


def sphere_dataset(n_cams=8, H=64, W=64, seed=0, radius=0.6, camera_angle_x=0.6911):
    """(poses [B,4,4], intrinsics (fx,fy,cx,cy), images [B,H,W,4]) of the analytic sphere scene: the arrays a
    Blender-style provider hands to the trainer (provider.py:269-281 intrinsics from camera_angle_x)."""
    poses = hemisphere_poses(n_cams, seed=seed)
    fl = W / (2 * np.tan(camera_angle_x / 2))
    pix = np.stack([np.repeat(np.arange(n_cams), H * W), np.tile(np.arange(H * W), n_cams)], -1)
    o, d = get_rays(poses, pix, H, W, camera_angle_x)
    images = sphere_scene_rgba(o, d, radius).reshape(n_cams, H, W, 4)
    return poses, (fl, fl, W / 2, H / 2), images


def detail_scene_rgba(rays_o, rays_d):
    """A second analytic scene, with FINE structure (the sphere scene's finest wavelet levels carry almost
    nothing): opaque solids with view-consistent 3D procedural albedo, first hit wins, alpha = 1 on a hit --
      * a ball (r 0.42, centre (-0.3, 0.05, 0)) with a 3D checker of 0.022-wide cells (136 cells across the 3.0-wide
        bound: 15 texels per cell at R = 2048, sharp cell edges at every scale),
      * a box [0.12, 0.62] x [-0.28, 0.30] x [-0.30, 0.28] with diagonal stripes of wavelength 0.018 (167 cycles across),
      * a THIN plate (thickness 0.012 = 8 texels) under both, [-0.75, 0.75] x [-0.42, -0.408] x [-0.75, 0.75], checker 0.03,
      * a thin vertical fin (thickness 0.010) x in [-0.02, -0.01], y in [-0.408, 0.35], z in [-0.5, 0.5], stripes 0.025.
    -> [N,4] like a Blender RGBA pixel."""
    o, d = np.asarray(rays_o, np.float64), np.asarray(rays_d, np.float64)
    N = o.shape[0]
    t_best = np.full(N, np.inf)
    rgb = np.zeros((N, 3))

    def take(t, hit, color_fn, normal):
        better = hit & (t < t_best) & (t > 0)
        if not better.any():
            return
        p = o[better] + t[better, None] * d[better]
        c = color_fn(p)
        n = normal(p) if callable(normal) else normal[better]
        shade = 0.75 + 0.25 * n[:, 1:2]                      # light from +y, view-independent
        rgb[better] = np.clip(c * shade, 0.0, 1.0)
        t_best[better] = t[better]

    def checker(cell, a, b):
        def f(p):
            k = np.floor(p / cell).astype(np.int64).sum(-1) & 1
            return np.where(k[:, None] == 0, np.asarray(a)[None], np.asarray(b)[None])
        return f

    def stripes(wl, direction, a, b):
        direction = np.asarray(direction, np.float64) / np.linalg.norm(direction)

        def f(p):
            k = np.floor((p @ direction) / (wl / 2)).astype(np.int64) & 1
            return np.where(k[:, None] == 0, np.asarray(a)[None], np.asarray(b)[None])
        return f

    # ball
    c0, r0 = np.array([-0.3, 0.05, 0.0]), 0.42
    oc = o - c0
    bq = (oc * d).sum(-1)
    disc = bq * bq - ((oc * oc).sum(-1) - r0 * r0)
    t = -bq - np.sqrt(np.clip(disc, 0, None))
    take(t, disc > 0, checker(0.022, (0.95, 0.35, 0.15), (0.15, 0.35, 0.9)), lambda p: (p - c0) / r0)

    def box(lo, hi, color_fn):
        lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / d
            t0, t1 = (lo - o) * inv, (hi - o) * inv
        tn, tf = np.minimum(t0, t1), np.maximum(t0, t1)
        tnear, tfar = tn.max(-1), tf.min(-1)
        hit = (tnear < tfar) & (tnear > 0)
        axis = tn.argmax(-1)
        nrm = np.zeros((N, 3))
        nrm[np.arange(N), axis] = -np.sign(d[np.arange(N), axis])
        take(tnear, hit, color_fn, nrm)

    box((0.12, -0.28, -0.30), (0.62, 0.30, 0.28), stripes(0.018, (1.0, 1.0, 0.6), (0.9, 0.85, 0.2), (0.1, 0.5, 0.25)))
    box((-0.75, -0.42, -0.75), (0.75, -0.408, 0.75), checker(0.03, (0.85, 0.85, 0.85), (0.2, 0.2, 0.25)))
    box((-0.02, -0.408, -0.5), (-0.01, 0.35, 0.5), stripes(0.025, (0.0, 1.0, 1.0), (0.9, 0.2, 0.6), (0.2, 0.8, 0.8)))
    hit = np.isfinite(t_best)
    return np.concatenate([rgb * hit[:, None], hit[:, None].astype(np.float64)], -1).astype(np.float32)


def detail_dataset(n_cams=8, H=64, W=64, seed=0, camera_angle_x=0.6911):
    """sphere_dataset's counterpart for detail_scene_rgba."""
    poses = hemisphere_poses(n_cams, seed=seed)
    fl = W / (2 * np.tan(camera_angle_x / 2))
    images = np.empty((n_cams, H, W, 4), np.float32)
    for c in range(n_cams):                                   # per camera: the float64 temporaries stay small
        pix = np.stack([np.full(H * W, c), np.arange(H * W)], -1)
        o, d = get_rays(poses, pix, H, W, camera_angle_x)
        images[c] = detail_scene_rgba(o, d).reshape(H, W, 4)
    return poses, (fl, fl, W / 2, H / 2), images


def init_field_parameters(model, seed=0):
    """SURVEY.md 8(d) field parameters: LL = 0.1*N(0,1); level-i coefficients ~ N(0, (0.02 * 2^-i)^2);
    Linear weights = torch default init under a fixed seed."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    enc = model.encoder
    with torch.no_grad():
        enc.planes_features.copy_((0.1 * torch.randn(enc.planes_features.shape, generator=g)).to(enc.planes_features.device))
        for i, p in enumerate(enc.planes_features_wavelet_coefs):
            dev_g = torch.Generator(device=p.device).manual_seed(seed + 1 + i)
            p.copy_(torch.randn(p.shape, generator=dev_g, device=p.device) * (0.02 * 2.0 ** (-i)))
        for lin in list(model.sigma_net) + list(model.color_net):
            w = (torch.rand(lin.weight.shape, generator=g) * 2 - 1) / float(np.sqrt(lin.weight.shape[1]))
            lin.weight.copy_(w.to(lin.weight.device))
