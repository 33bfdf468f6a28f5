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
