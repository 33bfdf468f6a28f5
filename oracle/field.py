"""CPU torch-fp32 restatement of the field and of one optimisation step (checker only).

TEST INFRASTRUCTURE ONLY (see oracle/trinerflet_oracle.c header): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by trinerflet_amd/.

Follows, with the reference's own torch operators where the reference uses torch:
  triplane lookup  reconstruction/triplaneencoder/triplane_encoder.py:293-332  (F.grid_sample, bilinear,
                   border, align_corners=True; planes (3,C,R,R); plane axes (x,z),(x,y),(y,z))
  field            reconstruction/nerf/network.py:118-147   (bias-free Linear x5, ReLU, trunc_exp, sigmoid)
  trunc_exp        reconstruction/activation.py:5-17
  SH-4             aux_libs/shencoder/src/shencoder.cu:50-68
  planes           triplane_encoder.py:364-405 with pytorch_wavelets.DWTInverse(mode='zero') restated as
                   depthwise conv_transpose2d (stride 2, padding L-2) -- pinned against PyWavelets through
                   oracle/trinerflet_oracle.c and tests/golden/idwt_pywt.npz (tests/test_oracle.py)
  loss / reg       reconstruction/nerf/utils.py:595,639-655
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import cref

PLANE_AXES = ((0, 2), (0, 1), (1, 2))  # (grid x, grid y) source coordinates per plane


class _TruncExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


def sh4(d):
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    return torch.stack([
        torch.full_like(x, 0.28209479177387814), -0.48860251190291987 * y, 0.48860251190291987 * z,
        -0.48860251190291987 * x, 1.0925484305920792 * xy, -1.0925484305920792 * yz,
        0.94617469575755997 * z2 - 0.31539156525251999, -1.0925484305920792 * xz,
        0.54627421529603959 * x2 - 0.54627421529603959 * y2, 0.59004358992664352 * y * (-3.0 * x2 + y2),
        2.8906114426405538 * xy * z, 0.45704579946446572 * y * (1.0 - 5.0 * z2),
        0.3731763325901154 * z * (5.0 * z2 - 3.0), 0.45704579946446572 * x * (1.0 - 5.0 * z2),
        1.4453057213202769 * z * (x2 - y2), 0.59004358992664352 * x * (-x2 + 3.0 * y2)], -1)


def triplane_features(planes, xyz, bound):
    """planes (3,C,R,R), xyz [N,3] -> [N,3C] via F.grid_sample exactly as sample_from_planes_aux."""
    u = xyz / bound
    grid = torch.stack([torch.stack([u[:, a], u[:, b]], -1) for a, b in PLANE_AXES], 0).unsqueeze(2)  # 3,N,1,2
    s = F.grid_sample(planes, grid, mode='bilinear', padding_mode='border', align_corners=True)  # 3,C,N,1
    return s.permute(2, 0, 1, 3).squeeze(-1).reshape(xyz.shape[0], -1)


class _RoundFp16(torch.autograd.Function):
    """Round the VALUE to fp16 with a straight-through gradient.  (Plain `x.half().float()` must not be used
    here: autograd of the dtype casts also rounds the GRADIENT to fp16, in unscaled units.)"""
    @staticmethod
    def forward(ctx, x):
        return x.half().float()

    @staticmethod
    def backward(ctx, g):
        return g


def round_fp16(x):
    return _RoundFp16.apply(x)


class _RoundGradFp16(torch.autograd.Function):
    """Identity whose backward rounds the incoming gradient to fp16: the backward GEMMs of an fp16 Linear
    (torch autocast in the reference, MFMA operands in the kernel) see their dY operand in half precision."""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.half().float()


def mlp(feats, dirs, W, fp16=False, grad_fp16=False):
    """W = [W0,W1,W2,W3,W4] in nn.Linear layout.  fp16=True rounds Linear inputs and weights to half (the
    kernel's MFMA operand precision) while accumulating in fp32; grad_fp16=True also rounds every dY."""
    def lin(x, w):
        if fp16:
            y = F.linear(round_fp16(x), round_fp16(w))
        else:
            y = F.linear(x, w)
        return _RoundGradFp16.apply(y) if grad_fp16 else y
    h = torch.relu(lin(feats, W[0]))
    o = lin(h, W[1])
    sigma = _TruncExp.apply(o[:, 0])
    z = torch.cat([sh4(dirs), o[:, 1:]], -1)
    h = torch.relu(lin(z, W[2]))
    h = torch.relu(lin(h, W[3]))
    rgb = torch.sigmoid(lin(h, W[4]))
    return sigma, rgb


def field(planes, xyz, dirs, W, bound, fp16=False, plane_half=False, grad_fp16=False):
    p = round_fp16(planes) if plane_half else planes
    return mlp(triplane_features(p, xyz, bound), dirs, W, fp16=fp16, grad_fp16=grad_fp16)


def synthesis_filters(wave):
    _, L, lo, hi = cref.wavelet_taps(wave)
    return L, torch.tensor(lo, dtype=torch.float32), torch.tensor(hi, dtype=torch.float32)


def idwt_level_torch(x, yh, wave):
    """One level of build_planes with torch conv_transpose2d (what pytorch_wavelets' SFB2D runs):
    x (3,C,n,n), yh (3,C,3,n,n) -> (3,C,2n,2n)."""
    L, g0, g1 = synthesis_filters(wave)
    pad = (L - 2) // 4
    C = x.shape[1]
    yl = F.pad(2 * x, (pad, pad, pad, pad))
    yhp = F.pad(yh, (pad, pad, pad, pad))
    lh, hl, hh = yhp[:, :, 0], yhp[:, :, 1], yhp[:, :, 2]

    def sfb1d(lo, hi, dim):
        shape = [1, 1, 1, 1]
        shape[dim] = L
        k0 = g0.to(lo.dtype).reshape(shape).repeat(C, 1, 1, 1)
        k1 = g1.to(lo.dtype).reshape(shape).repeat(C, 1, 1, 1)
        s = (2, 1) if dim == 2 else (1, 2)
        p = (L - 2, 0) if dim == 2 else (0, L - 2)
        return F.conv_transpose2d(lo, k0, stride=s, padding=p, groups=C) + \
            F.conv_transpose2d(hi, k1, stride=s, padding=p, groups=C)
    lo = sfb1d(yl, lh, 2)
    hi = sfb1d(hl, hh, 2)
    return sfb1d(lo, hi, 3)


def build_planes_torch(ll, coefs, wave):
    x = ll
    for yh in coefs:
        x = idwt_level_torch(x, yh, wave)
    return x


def composite_train_torch(sigmas, rgbs, deltas, rays, T_thresh=1e-4):
    """Differentiable torch restatement of composite_rays_train (raymarching.cu:501-577) for small cases:
    per ray sequential recurrence (the early stop included)."""
    N = rays.shape[0]
    ws, depth, image = [], [], []
    M = sigmas.shape[0]
    for n in range(N):
        idx, off, cnt = int(rays[n, 0]), int(rays[n, 1]), int(rays[n, 2])
        T = torch.ones((), dtype=sigmas.dtype)
        r = torch.zeros(3, dtype=sigmas.dtype)
        w_sum = torch.zeros((), dtype=sigmas.dtype)
        d = torch.zeros((), dtype=sigmas.dtype)
        t = torch.zeros((), dtype=sigmas.dtype)
        if cnt > 0 and off + cnt <= M:
            for s in range(off, off + cnt):
                alpha = 1 - torch.exp(-sigmas[s] * deltas[s, 0])
                w = alpha * T
                r = r + w * rgbs[s]
                t = t + deltas[s, 1]
                d = d + w * t
                w_sum = w_sum + w
                T = T * (1 - alpha)
                if float(T) < T_thresh:
                    break
        ws.append((idx, w_sum)); depth.append((idx, d)); image.append((idx, r))
    out_ws = torch.zeros(N, dtype=sigmas.dtype)
    out_d = torch.zeros(N, dtype=sigmas.dtype)
    out_i = torch.zeros(N, 3, dtype=sigmas.dtype)
    ws_l, d_l, i_l = [None] * N, [None] * N, [None] * N
    for (i, v) in ws: ws_l[i] = v
    for (i, v) in depth: d_l[i] = v
    for (i, v) in image: i_l[i] = v
    return torch.stack(ws_l), torch.stack(d_l), torch.stack(i_l)


def wavelet_reg(coefs, lam):
    """utils.py:639-655: lam * (1/J) * sum_l mean|c_l| * numel_l / sum numel."""
    total = sum(c.numel() for c in coefs)
    return lam * sum(c.abs().mean() * (c.numel() / total) for c in coefs) / len(coefs)


def lr_factor(it, iters, warmup_steps, sched_base=0.1, warmup_factor=1e-3, sched_exp=2.5):
    """decay_function, utils.py:55-62 (accumelate_steps = 1)."""
    w = max(warmup_steps, 0)
    if it < w:
        return sched_base * warmup_factor + it * (1 - warmup_factor) / (w - 1)
    return sched_base ** (min((it - w) / iters, 1) ** sched_exp)
