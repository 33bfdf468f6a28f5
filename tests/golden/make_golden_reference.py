"""Generates tests/golden/triplane_reference.npz by RUNNING THE REFERENCE's own Python in this
container (never on the GPU box; /root/reference does not travel):

    python tests/golden/make_golden_reference.py

What runs: reconstruction/triplaneencoder/triplane_encoder.py::TriPlaneVolume, imported from
/root/reference unmodified.  Its third-party dependency `pytorch_wavelets` (1.3.0, un-vendored,
requirements2.txt:113) is not installed here; the module registered below is an ADAPTER, not an
implementation: DWTForward/DWTInverse forward every call to the real PyWavelets (`pywt.dwt2` /
`pywt.idwt2`, mode='zero') running under /opt/conda/bin/python3.9 -- the library pytorch_wavelets
itself wraps.  No wavelet arithmetic is restated here.

Stored vectors (all float64 unless noted):
  idwt/<wave>/{ll, coef0, coef1, planes}     TriPlaneVolume.get_planes() for each supported wavelet
  sample/{planes, xyz, bound, feats, cot, dplanes}
        TriPlaneVolume.forward(xyz, bound) on 257 points (corners, out-of-range, random) and the
        autograd VJP of a random cotangent w.r.t. the planes (float32 torch, as the reference runs)
  trunc_exp/{x, y, g, gx}                     activation.py forward/backward
"""
import os
import subprocess
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/reconstruction"
PY39 = "/opt/conda/bin/python3.9"

_WORKER = r"""
import sys, numpy as np, pywt
op, wave, path = sys.argv[1:4]
d = np.load(path)
if op == 'idwt2':
    yl, yh = d['yl'], d['yh']          # [B,C,h,w], [B,C,3,h,w]
    out = np.stack([np.stack([pywt.idwt2((yl[b,c], (yh[b,c,0], yh[b,c,1], yh[b,c,2])), wave, mode='zero')
                              for c in range(yl.shape[1])]) for b in range(yl.shape[0])])
    np.save(path + '.out.npy', out)
else:
    x = d['x']
    ll, hs = [], []
    for b in range(x.shape[0]):
        l_, h_ = [], []
        for c in range(x.shape[1]):
            a, (h, v, dd) = pywt.dwt2(x[b,c], wave, mode='zero')
            l_.append(a); h_.append(np.stack([h, v, dd]))
        ll.append(np.stack(l_)); hs.append(np.stack(h_))
    np.savez(path + '.out.npz', yl=np.stack(ll), yh=np.stack(hs))
"""


def _call_pywt(op, wave, **arrays):
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "in.npz")
        np.savez(path, **arrays)
        subprocess.check_call([PY39, "-W", "ignore", "-c", _WORKER, op, wave, path])
        if op == "idwt2":
            return np.load(path + ".out.npy")
        d = np.load(path + ".out.npz")
        return d["yl"], d["yh"]


class DWTForward(torch.nn.Module):
    def __init__(self, J=1, wave="db1", mode="zero"):
        super().__init__()
        assert J == 1 and mode == "zero"
        self.wave = wave

    def forward(self, x):
        yl, yh = _call_pywt("dwt2", self.wave, x=x.detach().double().numpy())
        return torch.from_numpy(yl).to(x.dtype), [torch.from_numpy(yh).to(x.dtype)]


class DWTInverse(torch.nn.Module):
    def __init__(self, wave="db1", mode="zero"):
        super().__init__()
        assert mode == "zero"
        self.wave = wave

    def forward(self, coeffs):
        yl, (yh,) = coeffs
        out = _call_pywt("idwt2", self.wave, yl=yl.detach().double().numpy(), yh=yh.detach().double().numpy())
        return torch.from_numpy(out).to(yl.dtype)


def main():
    adapter = types.ModuleType("pytorch_wavelets")
    adapter.DWTForward, adapter.DWTInverse = DWTForward, DWTInverse
    sys.modules["pytorch_wavelets"] = adapter
    sys.path.insert(0, REF)
    from triplaneencoder.triplane_encoder import TriPlaneVolume  # the reference class, unmodified
    from activation import trunc_exp

    torch.manual_seed(0)
    torch.set_default_dtype(torch.float64)
    out = {}
    for wave in ("haar", "bior2.2", "bior4.4", "bior2.6", "bior6.8"):
        vol = TriPlaneVolume(number_of_features=2, plane_resolution=32, inner_multi_res_scale=4,
                             wavelet_type=wave)
        with torch.no_grad():
            for i, p in enumerate(vol.planes_features_wavelet_coefs):
                p.copy_(torch.randn_like(p) * 0.3)
        assert [tuple(p.shape) for p in vol.planes_features_wavelet_coefs] == [(3, 2, 3, 8, 8), (3, 2, 3, 16, 16)]
        assert tuple(vol.planes_features.shape) == (3, 2, 8, 8)
        planes = vol.get_planes()
        out[f"idwt/{wave}/ll"] = vol.planes_features.detach().numpy()
        out[f"idwt/{wave}/coef0"] = vol.planes_features_wavelet_coefs[0].detach().numpy()
        out[f"idwt/{wave}/coef1"] = vol.planes_features_wavelet_coefs[1].detach().numpy()
        out[f"idwt/{wave}/planes"] = planes.detach().numpy()
        print(wave, "planes", tuple(planes.shape))

    # --- sampling: plain (non-wavelet) planes so that autograd reaches them through grid_sample
    torch.set_default_dtype(torch.float32)
    # (the reference's inner_multi_res_scale=1 mode is broken at triplane_encoder.py:412, so the planes are
    #  handed to sample_from_planes explicitly; forward() = sample_from_planes(...).view(N,-1), :523-526)
    vol = TriPlaneVolume(number_of_features=4, plane_resolution=32, inner_multi_res_scale=4, wavelet_type="haar")
    planes_leaf = (0.5 * torch.randn(3, 4, 32, 32, generator=torch.Generator().manual_seed(7))).requires_grad_(True)
    bound = 1.5
    g = torch.Generator().manual_seed(1)
    xyz = (torch.rand(257, 3, generator=g) * 2 - 1) * bound
    corners = torch.tensor([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=torch.float32) * bound
    xyz[:8] = corners
    xyz[8:16] = corners * 1.25          # outside the box: border padding
    xyz[16] = torch.zeros(3)
    xyz[17] = torch.tensor([bound, 0.0, -bound])
    feats = vol.sample_from_planes(xyz, plane_features=planes_leaf, lbound=bound).view(xyz.shape[0], -1)
    vol.last_used_planes = planes_leaf.detach()          # forward() reads the cache (:408-409)
    assert torch.equal(vol(xyz, bound), feats.detach())  # TriPlaneVolume.forward -> [N, 3C]
    cot = torch.randn(feats.shape, generator=g)
    (dpl,) = torch.autograd.grad(feats, planes_leaf, cot)
    out["sample/planes"] = planes_leaf.detach().numpy()
    out["sample/xyz"] = xyz.numpy()
    out["sample/bound"] = np.array(bound)
    out["sample/feats"] = feats.detach().numpy()
    out["sample/cot"] = cot.numpy()
    out["sample/dplanes"] = dpl.numpy()
    assert feats.shape == (257, 12) and vol.output_dim == 12

    x = torch.linspace(-20, 20, 41, requires_grad=True)
    y = trunc_exp(x)
    gy = torch.linspace(0.5, 1.5, 41)
    (gx,) = torch.autograd.grad(y, x, gy)
    out["trunc_exp/x"], out["trunc_exp/y"] = x.detach().numpy(), y.detach().numpy()
    out["trunc_exp/g"], out["trunc_exp/gx"] = gy.numpy(), gx.numpy()

    np.savez_compressed(os.path.join(HERE, "triplane_reference.npz"), **out)
    print("wrote triplane_reference.npz")


if __name__ == "__main__":
    main()
