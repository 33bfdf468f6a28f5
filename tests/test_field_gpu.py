"""GPU parity of the fused field kernels (lookup + sigma MLP + SH + colour MLP, forward and backward)
against the torch-fp32 restatement of NeRFNetwork.forward (oracle/field.py)."""
import numpy as np
import pytest
import torch

from oracle import field as ofield

pytestmark = pytest.mark.gpu


def _make(C, H, R, M, seed=0, bound=1.5):
    g = torch.Generator().manual_seed(seed)
    planes = torch.randn(3, C, R, R, generator=g) * 0.5
    xyz = (torch.rand(M, 3, generator=g) * 2 - 1) * bound
    dirs = torch.randn(M, 3, generator=g)
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    shapes = [(H, 3 * C), (16, H), (H, 31), (H, H), (3, H)]
    W = [(torch.rand(s, generator=g) * 2 - 1) / np.sqrt(s[1]) for s in shapes]  # nn.Linear default init range
    return planes, xyz, dirs, W, bound


def _net(cuda, C, H, R, plane_dtype):
    from trinerflet_amd.nerf.network import NeRFNetwork
    net = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, hidden_dim=H, hidden_dim_color=H,
                      triplane_channels=C, triplane_resolution=R, triplane_wavelet_levels=1, plane_dtype=plane_dtype)
    return net.to(cuda)


@pytest.mark.parametrize("C,H,R,M", [(16, 64, 64, 5000), (32, 64, 128, 4097), (48, 128, 64, 1500)])
@pytest.mark.parametrize("plane_dtype", [torch.float32, torch.float16])
def test_field_forward(cuda, C, H, R, M, plane_dtype):
    from trinerflet_amd.nerf import field as gfield
    from trinerflet_amd.triplaneencoder.triplane_encoder import _ToTexelMajor
    planes, xyz, dirs, W, bound = _make(C, H, R, M)
    half = plane_dtype == torch.float16
    tm = _ToTexelMajor.apply(planes.to(cuda), half)
    Wg = [w.to(cuda) for w in W]
    sigma, rgb = gfield.fused_field(tm, xyz.to(cuda), dirs.to(cuda), *Wg, bound)
    # (a) oracle with the kernel's operand precision (fp16 MFMA inputs, fp32 accumulate)
    s16, c16 = ofield.field(planes, xyz, dirs, W, bound, fp16=True, plane_half=half)
    np.testing.assert_allclose(sigma.cpu().numpy(), s16.numpy(), rtol=3e-3, atol=1e-6)
    np.testing.assert_allclose(rgb.cpu().numpy(), c16.numpy(), rtol=0, atol=1e-3)
    # (b) plain fp32 oracle: BASELINE.json's tolerance "RGB/sigma within 1e-3 (fp16)"; sigma is exp(logit),
    #     so the bound is relative and scales with |logit| * 2^-11
    s32, c32 = ofield.field(planes, xyz, dirs, W, bound)
    np.testing.assert_allclose(rgb.cpu().numpy(), c32.numpy(), rtol=0, atol=3e-3)
    rel = np.abs(sigma.cpu().numpy() - s32.numpy()) / s32.numpy()
    assert np.median(rel) < 1e-3 and rel.max() < 2e-2


def test_density_matches_forward(cuda):
    C, H, R, M = 32, 64, 64, 3000
    planes, xyz, dirs, W, bound = _make(C, H, R, M, seed=3)
    net = _net(cuda, C, H, R, torch.float16)
    with torch.no_grad():
        net.encoder.planes_features.copy_(planes.to(cuda))
        for lin, w in zip(list(net.sigma_net) + list(net.color_net), W):
            lin.weight.copy_(w.to(cuda))
        s, c = net(xyz.to(cuda), dirs.to(cuda))
        d = net.density(xyz.to(cuda))
    assert torch.equal(d['sigma'], s)
    feats = ofield.triplane_features(planes.half().float(), xyz, bound)
    o = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(feats.half().float(), W[0].half().float())).half().float(),
                                   W[1].half().float())
    np.testing.assert_allclose(d['geo_feat'].cpu().numpy(), o[:, 1:].numpy(), rtol=0, atol=2e-3)


def _relerr(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("C,H,R,M", [(16, 64, 64, 3000), (32, 64, 96, 4133), (48, 128, 32, 700)])
def test_field_backward(cuda, C, H, R, M):
    """d(sigma,rgb)/d(planes, W0..W4) of the fused kernel vs torch autograd of the fp32 restatement.
    The kernel rounds MFMA operands (activations, weights and incoming gradients) to fp16, so the comparison
    is norm-wise: relative L2 error of every gradient tensor below 5e-3."""
    from trinerflet_amd.nerf import field as gfield
    from trinerflet_amd.triplaneencoder.triplane_encoder import _ToTexelMajor
    planes, xyz, dirs, W, bound = _make(C, H, R, M, seed=11)
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, generator=g)
    b = torch.randn(M, 3, generator=g)
    # oracle (fp16-operand emulation keeps the forward identical to the kernel; grads flow in fp32)
    pl_o = planes.clone().requires_grad_(True)
    W_o = [w.clone().requires_grad_(True) for w in W]
    s_o, c_o = ofield.field(pl_o, xyz, dirs, W_o, bound, fp16=True, plane_half=False)
    ((s_o * a).sum() + (c_o * b).sum()).backward()
    # kernel
    pl_g = planes.to(cuda).requires_grad_(True)
    W_g = [w.to(cuda).requires_grad_(True) for w in W]
    tm = _ToTexelMajor.apply(pl_g, False)
    s_g, c_g = gfield.fused_field(tm, xyz.to(cuda), dirs.to(cuda), *W_g, bound)
    ((s_g * a.to(cuda)).sum() + (c_g * b.to(cuda)).sum()).backward()
    for k, (wg, wo) in enumerate(zip(W_g, W_o)):
        e = _relerr(wg.grad.cpu().numpy(), wo.grad.numpy().astype(np.float64))
        assert e < 5e-3, f"dW{k} rel err {e}"
    e = _relerr(pl_g.grad.cpu().numpy(), pl_o.grad.numpy().astype(np.float64))
    assert e < 5e-3, f"dplanes rel err {e}"
    # untouched texels must stay exactly zero
    untouched = (pl_o.grad == 0)
    assert float(pl_g.grad.cpu()[untouched].abs().max()) == 0.0


def test_network_fused_equals_modular(cuda):
    """NeRFNetwork.forward: fused kernel vs the modular GPU path (HIP lookup + HIP SH + torch Linear, fp32)."""
    C, H, R, M = 32, 64, 64, 2000
    planes, xyz, dirs, W, bound = _make(C, H, R, M, seed=4)
    net = _net(cuda, C, H, R, torch.float32)
    with torch.no_grad():
        net.encoder.planes_features.copy_(planes.to(cuda))
        for lin, w in zip(list(net.sigma_net) + list(net.color_net), W):
            lin.weight.copy_(w.to(cuda))
        s_f, c_f = net(xyz.to(cuda), dirs.to(cuda))
        net.force_modular = True
        s_m, c_m = net(xyz.to(cuda), dirs.to(cuda))
    np.testing.assert_allclose(c_f.cpu().numpy(), c_m.cpu().numpy(), rtol=0, atol=3e-3)
    rel = (s_f - s_m).abs() / s_m
    assert float(rel.median()) < 1e-3 and float(rel.max()) < 2e-2


@pytest.mark.parametrize("C,H,R,M", [(32, 64, 96, 6000), (16, 64, 64, 2500), (48, 128, 64, 5003)])
def test_binned_plane_gradient_equals_atomic(cuda, C, H, R, M):
    """TrainStep's atomic-free path (dF as fp16 -> tile-sorted LDS accumulation) against the atomic scatter
    and the oracle; includes border / out-of-range samples (clamped footprints) and a device-side row count."""
    from trinerflet_amd.nerf import field as gfield
    from trinerflet_amd.triplaneencoder.triplane_encoder import _ToTexelMajor
    planes, xyz, dirs, W, bound = _make(C, H, R, M, seed=21)
    xyz[:64] *= 1.2                      # outside the box: border clamp
    xyz[64:80] = torch.tensor([bound, -bound, bound])
    g = torch.Generator().manual_seed(9)
    a, b = torch.randn(M, generator=g), torch.randn(M, 3, generator=g)
    tm = _ToTexelMajor.apply(planes.to(cuda), True)
    packed = gfield.pack_weights(*[w.to(cuda) for w in W], C, H)
    xg, dg = xyz.to(cuda), dirs.to(cuda)
    m_act = torch.tensor([M - 37, 0], dtype=torch.int32, device=cuda)  # rows past the count are ignored
    s, c, feats = gfield.field_forward(tm, xg, dg, packed, bound, C, R, H, save_feats=True, m_actual=m_act)
    nW = sum(w.numel() for w in W)
    g_at, w_at = torch.zeros(3, R, R, C, device=cuda), torch.zeros(nW, device=cuda)
    gfield.field_backward(a.to(cuda), b.to(cuda), None, None, feats, xg, dg, packed, bound, C, R, H, g_at, w_at,
                          m_actual=m_act)
    g_bin = torch.full((3, R, R, C), float("nan"), device=cuda)      # every tile must be overwritten
    w_bin = torch.zeros(nW, device=cuda)
    dfeat = torch.empty(3, M, C, dtype=torch.float16, device=cuda)      # plane-major [3,M,C]
    if H > 64:   # the two-launch backward of the hidden-128 network reads sigma and says so when it is missing
        with pytest.raises(RuntimeError):
            gfield.field_backward(a.to(cuda), b.to(cuda), None, None, feats, xg, dg, packed, bound, C, R, H, g_bin,
                                  w_bin, m_actual=m_act, dfeat=dfeat)
        w_bin.zero_()
    gfield.field_backward(a.to(cuda), b.to(cuda), s, None, feats, xg, dg, packed, bound, C, R, H, g_bin, w_bin,
                          m_actual=m_act, dfeat=dfeat)
    gfield.plane_grad_binned(dfeat, xg, bound, C, R, g_bin, m_actual=m_act)
    assert torch.isfinite(g_bin).all()
    g_cm = torch.full((3, C, R, R), float("nan"), device=cuda)      # the (3,C,R,R) output the adjoint IDWT reads
    gfield.plane_grad_binned(dfeat, xg, bound, C, R, g_cm, m_actual=m_act, channel_major=True)
    assert _relerr(g_cm.permute(0, 2, 3, 1).cpu().numpy(), g_bin.cpu().numpy().astype(np.float64)) < 1e-6
    assert torch.equal(w_bin, w_at) or _relerr(w_bin.cpu().numpy(), w_at.cpu().numpy()) < 1e-6
    assert _relerr(g_bin.cpu().numpy(), g_at.cpu().numpy().astype(np.float64)) < 1e-3   # fp16 rounding of dF
    assert torch.equal(g_bin == 0, g_at == 0)                                            # same support
    # oracle: the same samples through torch autograd
    n = M - 37
    pl = ofield.round_fp16(planes).requires_grad_(True)
    so, co = ofield.field(pl, xyz[:n], dirs[:n], W, bound, fp16=True)
    ((so * a[:n]).sum() + (co * b[:n]).sum()).backward()
    ref = pl.grad.permute(0, 2, 3, 1).numpy().astype(np.float64)
    assert _relerr(g_bin.cpu().numpy(), ref) < 5e-3


def test_tile_reduction_refuses_a_workspace_sorted_with_another_capacity(cuda):
    """The tile lists' positions lie behind 12 * M entry slots: sort and reduce must be given the same capacity M
    (ADVICE r03).  The library notes what a workspace was sorted with and refuses a mismatch."""
    from trinerflet_amd.nerf import field as gfield
    C, R, M = 16, 64, 2000
    g = torch.Generator().manual_seed(0)
    xyz = ((torch.rand(M, 3, generator=g) * 2 - 1) * 1.4).to(cuda)
    dfeat = torch.zeros(3, M, C, dtype=torch.float16, device=cuda)
    ws = gfield.plane_grad_sort(xyz, 1.5, R)
    out = torch.empty(3, R, R, C, device=cuda)
    gfield.plane_grad_reduce(ws, dfeat, xyz, 1.5, C, R, out)
    with pytest.raises(RuntimeError):
        gfield.plane_grad_reduce(ws, dfeat[:, :M - 128].contiguous(), xyz[:M - 128], 1.5, C, R, out)
