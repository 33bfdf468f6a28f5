"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads without a GPU and exports
every symbol include/trinerflet_hip.h declares; the Python mirrors expose the reference's names; the product
path refuses to run without the library or without a HIP device (no CPU fallback)."""
import ctypes
import inspect
import os

import pytest
import torch


def test_library_exports_every_declared_symbol():
    from trinerflet_amd import _lib, build
    lib = build.build()
    h = ctypes.CDLL(lib)
    declared = _lib.declared_symbols()
    assert len(declared) >= 27
    missing = [s for s in declared if not hasattr(h, s)]
    assert not missing, missing
    assert h.tnl_abi_version() == 1
    h.tnl_march_rays_train_workspace.restype = ctypes.c_uint32
    assert h.tnl_march_rays_train_workspace(ctypes.c_uint32(60000)) >= 60000 + 235
    h.tnl_field_packed_bytes.restype = ctypes.c_uint32
    assert h.tnl_field_packed_bytes(32, 64, 64) == 60 * 1024       # 32 forward + 28 transposed fragments
    assert h.tnl_field_packed_bytes(32, 64, 128) == 0              # unsupported shape -> 0, not a crash
    # sizes that pass 4 GB: an untrained occupancy grid lets 60 000 rays take 26 M samples (5 GB of saved features)
    h.tnl_field_feats_save_bytes.restype = ctypes.c_uint64
    assert h.tnl_field_feats_save_bytes(ctypes.c_uint32(26295552), 32, 64) == 26295552 * 96 * 2 > 2 ** 32
    assert h.tnl_field_feats_save_bytes(ctypes.c_uint32(1000), 48, 128) == (1024 * 144 + 1000 * 16) * 2   # rows in 32-sample tiles
    h.tnl_field_backward_workspace.restype = ctypes.c_uint64
    assert h.tnl_field_backward_workspace(ctypes.c_uint32(150_000_000), 48, 128, 128) > 2 ** 32
    h.tnl_march_rays_train_workspace_rec.restype = ctypes.c_uint32
    assert h.tnl_march_rays_train_workspace_rec(ctypes.c_uint32(60000), 1024) > 60000 * 1024
    assert h.tnl_march_rays_train_workspace_rec(ctypes.c_uint32(1 << 20), 4096) == 0   # does not fit: two-march form


def test_reference_api_surface():
    """Names and argument order of the reference modules."""
    from trinerflet_amd import raymarching
    want = {
        "near_far_from_aabb": ["rays_o", "rays_d", "aabb", "min_near"],
        "morton3D": ["coords"], "morton3D_invert": ["indices"], "packbits": ["grid", "thresh", "bitfield"],
        "march_rays_train": ["rays_o", "rays_d", "bound", "density_bitfield", "C", "H", "nears", "fars", "step_counter",
                             "mean_count", "perturb", "align", "force_all_rays", "dt_gamma", "max_steps"],
        "composite_rays_train": ["sigmas", "rgbs", "deltas", "rays", "T_thresh"],
        "march_rays": ["n_alive", "n_step", "rays_alive", "rays_t", "rays_o", "rays_d", "bound", "density_bitfield", "C",
                       "H", "near", "far", "align", "perturb", "dt_gamma", "max_steps"],
        "composite_rays": ["n_alive", "n_step", "rays_alive", "rays_t", "sigmas", "rgbs", "deltas", "weights_sum",
                           "depth", "image", "T_thresh"],
        "sph_from_ray": ["rays_o", "rays_d", "radius"],
    }
    for name, args in want.items():
        fn = getattr(raymarching, name)
        cls = fn.__self__
        got = list(inspect.signature(cls.forward).parameters)[1:]
        assert got[:len(args)] == args, (name, got)
    from trinerflet_amd.shencoder import SHEncoder
    assert SHEncoder(3, 4).output_dim == 16
    from trinerflet_amd.triplaneencoder.triplane_encoder import TriPlaneVolume
    vol = TriPlaneVolume(number_of_features=16, plane_resolution=512, inner_multi_res_scale=8)
    keys = set(vol.state_dict())
    assert {"planes_features", "plane_axes", "plane_normals", "planes_features_wavelet_coefs.0",
            "planes_features_wavelet_coefs.2", "idwt.g0_col", "idwt.g1_row"} <= keys
    assert [tuple(p.shape) for p in vol.planes_features_wavelet_coefs] == [(3, 16, 3, 64, 64), (3, 16, 3, 128, 128),
                                                                          (3, 16, 3, 256, 256)]
    assert tuple(vol.planes_features.shape) == (3, 16, 64, 64) and vol.output_dim == 48
    for m in ("forward", "get_planes", "reset_cahce", "get_wavelet_features", "get_wavelet_features_upscaled",
              "get_lbound_scale", "sample_from_planes", "get_params"):
        assert hasattr(vol, m)
    for m in ("get_grid_features", "get_params2", "init_upscale", "sample_from_planes_aux",
              "sample_from_planes_aux_rotation", "build_planes"):
        assert hasattr(vol, m)
    rot = TriPlaneVolume(number_of_features=16, plane_resolution=512, inner_multi_res_scale=8, learn_rotation_axis=True)
    assert tuple(rot.rotation_matrix.shape) == (16, 3, 3) and not rot.is_plain() and vol.is_plain()
    # wavelet_base_resolution: the analysis sizes at or below it stay uncropped (triplane_encoder.py:190-196)
    wbr = TriPlaneVolume(number_of_features=2, plane_resolution=128, inner_multi_res_scale=8, wavelet_base_resolution=41)
    assert [p.shape[-1] for p in wbr.planes_features_wavelet_coefs] == [28, 40, 64] and wbr.planes_features.shape[-1] == 28
    from trinerflet_amd.nerf.network import NeRFNetwork
    net = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, triplane_channels=16,
                      triplane_resolution=256, triplane_wavelet_levels=4, density_thresh=10)
    sd = set(net.state_dict())
    assert {"sigma_net.0.weight", "sigma_net.1.weight", "color_net.0.weight", "color_net.2.weight", "aabb_train",
            "aabb_infer", "density_grid", "density_bitfield", "step_counter", "encoder.planes_features"} <= sd
    assert net.cascade == 2 and net.grid_size == 128 and net.density_bitfield.numel() == 2 * 128 ** 3 // 8
    assert tuple(net.color_net[0].weight.shape) == (64, 31) and tuple(net.sigma_net[1].weight.shape) == (16, 64)
    import trinerflet_amd
    trinerflet_amd.install_dropin()
    import raymarching as rm  # noqa: F401  (what reconstruction/nerf/renderer.py:9 imports)
    from shencoder import SHEncoder as S2  # noqa: F401
    from triplaneencoder.triplane_encoder import TriPlaneVolume as T2  # noqa: F401
    from encoding import get_encoder  # noqa: F401
    assert rm.march_rays_train is raymarching.march_rays_train


def test_no_cpu_fallback():
    """The hot path must fail loudly off-GPU: wrappers refuse CPU tensors, and a missing library raises."""
    from trinerflet_amd import _lib
    from trinerflet_amd.shencoder import SHEncoder
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            SHEncoder(3, 4)(torch.randn(4, 3))
    saved = _lib.LIB_PATH, _lib._lib
    try:
        _lib.LIB_PATH, _lib._lib = os.path.join(os.path.dirname(saved[0]), "does_not_exist.so"), None
        with pytest.raises(_lib.HipLibraryMissing):
            _lib.lib()
    finally:
        _lib.LIB_PATH, _lib._lib = saved


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under trinerflet_amd/ or bench.py's product leg may import it."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for dp, _, files in os.walk(os.path.join(root, "trinerflet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b|oracle/_build|trinerflet_oracle", text, re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_ray_pool_has_no_cpu_path():
    import numpy as np
    from trinerflet_amd.raypool import RayPool
    pool = RayPool(np.tile(np.eye(4, dtype=np.float32), (1, 1, 1)), (4.0, 4.0, 2.0, 2.0), 4, 4,
                   np.zeros((1, 4, 4, 3), np.float32), device="cpu")
    with pytest.raises(RuntimeError):
        pool.batch(0, 8)


def test_eval_render_mode_selection():
    """render(..., render_mode=...) names the form of run_cuda's eval branch explicitly; the older keywords still select
    one when it is absent (passing an unrelated loop knob must not silently switch the path)."""
    import torch
    from trinerflet_amd.nerf.network import NeRFNetwork
    m = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, hidden_dim=64, hidden_dim_color=64,
                    triplane_channels=16, triplane_resolution=64, triplane_wavelet_levels=1)
    with torch.no_grad():
        assert m._eval_render_mode({}) == "kernel"
        assert m._eval_render_mode({"render_mode": "device_loop", "infer_min_step": 8}) == "device_loop"
        assert m._eval_render_mode({"render_mode": "host_loop"}) == "host_loop"
        assert m._eval_render_mode({"fused_render": True}) == "kernel"
        assert m._eval_render_mode({"fused_render": False}) == "device_loop"
        assert m._eval_render_mode({"device_loop": False}) == "host_loop"
        assert m._eval_render_mode({"infer_min_step": 4}) == "device_loop"
        with pytest.raises(ValueError):
            m._eval_render_mode({"render_mode": "kernel", "infer_min_step": 8})
        with pytest.raises(ValueError):
            m._eval_render_mode({"render_mode": "loop"})
    assert m._eval_render_mode({}) == "host_loop"          # under autograd only the reference's loop differentiates
    m.force_modular = True
    with torch.no_grad():
        assert m._eval_render_mode({"render_mode": "kernel"}) == "host_loop"   # no fused field for this configuration
