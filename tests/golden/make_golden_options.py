"""Generates tests/golden/triplane_options_reference.npz by RUNNING THE REFERENCE's TriPlaneVolume (imported from
/root/reference unmodified, `pytorch_wavelets` served by the PyWavelets adapter of make_golden_reference.py) with the
constructor options no README configuration uses:

    python tests/golden/make_golden_options.py

  tanh/        apply_activation_on_features=True                      -> planes, forward
  lbound/      lbound_auto_scale=True (lbound_scale set)              -> forward, VJP w.r.t. lbound_scale
  rot/         learn_rotation_axis=True (rotation_matrix seeded)      -> forward, VJP w.r.t. rotation_matrix
  up/          upscale_ratio_bound=0.5, upscale_levels=2              -> nested planes, forward
  cur/         inner_multi_res_scale_current=2                        -> learnable shapes, planes
  partial/     get_planes(max_res / max_scale / get_all_resolutions)  -> shapes and values
  grid/        get_grid_features(4)                                   -> lbound, features, grid
  wbr22/ wbr68/ wavelet_base_resolution > 0 (bior2.2 at 64^2 / 8, bior6.8 at 128^2 / 8) -> level sizes, planes, forward
All float32 (as the reference runs), C=2, R=32, scale 4, bior2.2.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_reference as G  # noqa: E402  (adapter classes + reference path)


def main():
    adapter = types.ModuleType("pytorch_wavelets")
    adapter.DWTForward, adapter.DWTInverse = G.DWTForward, G.DWTInverse
    sys.modules["pytorch_wavelets"] = adapter
    sys.path.insert(0, G.REF)
    from triplaneencoder.triplane_encoder import TriPlaneVolume

    out = {}
    C, R, scale, wave, bound = 2, 32, 4, "bior2.2", 1.5
    gen = torch.Generator().manual_seed(3)
    xyz = (torch.rand(200, 3, generator=gen) * 2 - 1) * bound
    xyz[:8] = torch.tensor([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=torch.float32) * bound
    xyz[8:40] *= 0.3                      # inside the nested zoom regions
    xyz[40:60] *= 0.6
    out["xyz"], out["bound"] = xyz.numpy(), np.array(bound)
    cot = torch.randn(200, 3 * C, generator=gen)
    out["cot"] = cot.numpy()

    def make(**kw):
        torch.manual_seed(11)
        vol = TriPlaneVolume(number_of_features=C, plane_resolution=R, inner_multi_res_scale=scale, wavelet_type=wave,
                             lbound=bound, **kw)
        with torch.no_grad():
            for p in vol.planes_features_wavelet_coefs:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.3)
        return vol

    def dump_params(tag, vol):
        out[f"{tag}/ll"] = vol.planes_features.detach().numpy()
        for i, p in enumerate(vol.planes_features_wavelet_coefs):
            out[f"{tag}/coef{i}"] = p.detach().numpy()

    # tanh
    vol = make(apply_activation_on_features=True)
    dump_params("tanh", vol)
    out["tanh/planes"] = vol.get_planes().detach().numpy()
    out["tanh/forward"] = vol(xyz, bound).detach().numpy()
    # lbound_auto_scale
    vol = make(lbound_auto_scale=True)
    with torch.no_grad():
        vol.lbound_scale.copy_(torch.tensor([0.3, -0.6, 0.9]))
    dump_params("lbound", vol)
    f = vol(xyz, bound)
    (g_s,) = torch.autograd.grad(f, [vol.lbound_scale], cot)   # (the adapter's IDWT is outside autograd: no VJP to LL)
    out["lbound/scale"] = vol.lbound_scale.detach().numpy()
    out["lbound/get_lbound_scale"] = vol.get_lbound_scale().detach().numpy()
    out["lbound/forward"], out["lbound/g_scale"] = f.detach().numpy(), g_s.numpy()
    out["lbound/param_names"] = np.array([n for n, _ in vol.named_parameters()])
    groups = vol.get_params2(0.01)
    out["lbound/params2_lrs"] = np.array([g["lr"] for g in groups])
    out["lbound/params2_sizes"] = np.array([len(g["params"]) for g in groups])
    # rotation
    vol = make(learn_rotation_axis=True)
    dump_params("rot", vol)
    out["rot/rotation_matrix"] = vol.rotation_matrix.detach().numpy()
    f = vol(xyz, bound)
    (g_r,) = torch.autograd.grad(f, [vol.rotation_matrix], cot)
    out["rot/forward"], out["rot/g_rot"] = f.detach().numpy(), g_r.numpy()
    # upscale
    vol = make(upscale_ratio_bound=0.5, upscale_levels=2)
    with torch.no_grad():
        for p in vol.upscale_wavelet_lst:
            p.copy_(torch.randn(p.shape, generator=gen) * 0.2)
    dump_params("up", vol)
    for i, p in enumerate(vol.upscale_wavelet_lst):
        out[f"up/wavelet{i}"] = p.detach().numpy()
    planes = vol.get_planes()
    for i, p in enumerate(planes):
        out[f"up/planes{i}"] = p.detach().numpy()
    out["up/base_resolution"] = np.array(vol.upscale_base_resolution_lst)
    out["up/base_corner"] = np.array(vol.upscale_base_corner_lst)
    out["up/bound_ratio"] = np.array(vol.upscale_bound_ratio_lst)
    f = vol(xyz, bound)
    out["up/forward"] = f.detach().numpy()
    out["up/n_upscaled_features"] = np.array(len(vol.get_wavelet_features_upscaled()))
    # inner_multi_res_scale_current
    vol = make(inner_multi_res_scale_current=2)
    dump_params("cur", vol)
    out["cur/n_learnable"] = np.array(len(vol.planes_features_wavelet_coefs))
    out["cur/planes"] = vol.get_planes().detach().numpy()
    # partial builds
    vol = make()
    dump_params("partial", vol)
    for tag, kw in (("max_res16", dict(max_res=16)), ("max_scale2", dict(max_scale=2)), ("full", dict())):
        vol.reset_cahce()
        out[f"partial/{tag}"] = vol.get_planes(**kw).detach().numpy()
    vol.reset_cahce()
    allres = vol.get_planes(get_all_resolutions=True)
    out["partial/all_n"] = np.array(len(allres))
    for i, a in enumerate(allres):
        out[f"partial/all{i}"] = a.detach().numpy()
    # grid features
    vol.reset_cahce()
    lb, feats, grid = vol.get_grid_features(4)
    out["grid/lbound"], out["grid/features"], out["grid/grid"] = np.array(lb), feats.detach().numpy(), grid.numpy()
    # wavelet_base_resolution: levels at or below it keep the uncropped analysis size and are synthesised without pad
    for tag, wv, res, sc, wbr in (("wbr22", "bior2.2", 64, 8, 12), ("wbr68", "bior6.8", 128, 8, 41)):
        torch.manual_seed(13)
        vol = TriPlaneVolume(number_of_features=C, plane_resolution=res, inner_multi_res_scale=sc, wavelet_type=wv,
                             lbound=bound, wavelet_base_resolution=wbr)
        with torch.no_grad():
            for p in vol.planes_features_wavelet_coefs:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.3)
        dump_params(tag, vol)
        out[f"{tag}/cfg"] = np.array([res, sc, wbr])
        out[f"{tag}/shapes"] = np.array([p.shape[-1] for p in vol.planes_features_wavelet_coefs] + [vol.planes_features.shape[-1]])
        out[f"{tag}/planes"] = vol.get_planes().detach().numpy()
        out[f"{tag}/forward"] = vol(xyz, bound).detach().numpy()
        print(tag, "level sizes", out[f"{tag}/shapes"], "planes", out[f"{tag}/planes"].shape)
    np.savez_compressed(os.path.join(HERE, "triplane_options_reference.npz"), **out)
    print("wrote", len(out), "arrays;", {k: v.shape for k, v in out.items() if k.startswith(("up/planes", "partial/", "cur/"))})


if __name__ == "__main__":
    main()
