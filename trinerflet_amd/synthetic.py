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
