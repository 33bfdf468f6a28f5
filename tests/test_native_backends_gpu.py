"""`_raymarching` / `_shencoder` -- the native module names the reference binds (aux_libs/raymarching/raymarching.py:9-12,
aux_libs/shencoder/sphere_harmonics.py:9-12) -- driven with the prototypes of raymarching.h:7-17 / shencoder.h the
way the reference's wrappers call them (caller allocates, zero-fills, passes sizes), against the frozen oracle outputs
of tests/golden/oracle_kernels.npz (F-SH, F-MARCH, F-COMP, F-INFER, F-GRID of SURVEY.md 8(c))."""
import hashlib
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_golden_oracle", os.path.join(HERE, "golden", "make_golden_oracle.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "oracle_kernels.npz"))


@pytest.fixture(scope="module")
def native(cuda):
    import trinerflet_amd
    trinerflet_amd.install_backends()
    import _raymarching
    import _shencoder
    assert "/trinerflet_amd/backends/" in _raymarching.__file__
    return _raymarching, _shencoder


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_sh_encode(cuda, fx, native):
    _, sh = native
    dirs = _t(fx["sh/dirs"], cuda)
    out = torch.empty(64, 16, device=cuda)
    sh.sh_encode_forward(dirs, out, 64, 3, 4, None)
    np.testing.assert_allclose(out.cpu().numpy(), fx["sh/out"], rtol=1e-6, atol=1e-7)
    with pytest.raises(RuntimeError):
        sh.sh_encode_forward(dirs.cpu(), out, 64, 3, 4, None)           # shencoder.cu:401-411 validates too


def test_grid_functions(cuda, fx, native):
    rm, _ = native
    ax = torch.arange(128, dtype=torch.int32, device=cuda)
    coords = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3).contiguous()
    codes = torch.empty(128 ** 3, dtype=torch.int32, device=cuda)
    rm.morton3D(coords, 128 ** 3, codes)
    assert np.array_equal(gen.sha(codes.cpu().numpy()), fx["grid/morton_sha"])
    back = torch.empty(128 ** 3, 3, dtype=torch.int32, device=cuda)
    rm.morton3D_invert(codes, 128 ** 3, back)
    assert torch.equal(back, coords)
    grid = np.random.default_rng(int(fx["grid/seed"])).standard_normal((2, 128 ** 3)).astype(np.float32)
    bits = torch.empty(2 * 128 ** 3 // 8, dtype=torch.uint8, device=cuda)
    rm.packbits(_t(grid, cuda), bits.numel(), float(fx["grid/thresh"]), bits)
    assert np.array_equal(gen.sha(bits.cpu().numpy()), fx["grid/packbits_sha"])


def _march(rm, fx, cfg, dev):
    """What _march_rays_train.forward (raymarching.py:161-233) does around the native call."""
    o, d = _t(fx["rays/o"], dev), _t(fx["rays/d"], dev)
    N = o.shape[0]
    aabb = torch.tensor([-gen.BOUND] * 3 + [gen.BOUND] * 3, device=dev)
    nears, fars = torch.empty(N, device=dev), torch.empty(N, device=dev)
    rm.near_far_from_aabb(o, d, aabb, N, 0.2, nears, fars)
    assert np.array_equal(nears.cpu().numpy(), fx["rays/nears"]) and np.array_equal(fars.cpu().numpy(), fx["rays/fars"])
    M = int(fx[f"march/{cfg}/M"])
    xyzs, dirs = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev)
    deltas = torch.zeros(M, 2, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    rm.march_rays_train(o, d, _t(fx["bitfield"], dev), gen.BOUND, 0.0, gen.MAX_STEPS, N, gen.CAS, gen.HG, M, nears, fars,
                        xyzs, dirs, deltas, rays, counter, _t(fx[f"march/{cfg}/noises"], dev))
    return xyzs, dirs, deltas, rays, counter, M


@pytest.mark.parametrize("cfg", ["plain", "perturb", "budget"])
def test_march_rays_train(cuda, fx, native, cfg):
    rm, _ = native
    xyzs, dirs, deltas, rays, counter, M = _march(rm, fx, cfg, cuda)
    assert np.array_equal(rays.cpu().numpy(), fx[f"march/{cfg}/rays"])              # ids, offsets, counts: exact
    assert np.array_equal(counter.cpu().numpy(), fx[f"march/{cfg}/counter"])
    m = min(int(counter[0]), M)
    rows = fx[f"march/{cfg}/sub_rows"]
    assert np.array_equal(xyzs.cpu().numpy()[rows], fx[f"march/{cfg}/sub_xyzs"])
    assert np.array_equal(deltas.cpu().numpy()[rows], fx[f"march/{cfg}/sub_deltas"])
    if cfg != "budget":
        # bit-exact positions / directions / step sizes of EVERY sample (hash of the bytes)
        assert np.array_equal(gen.sha(xyzs[:m].cpu().numpy()), fx[f"march/{cfg}/sha_xyzs"])
        assert np.array_equal(gen.sha(dirs[:m].cpu().numpy()), fx[f"march/{cfg}/sha_dirs"])
        assert np.array_equal(gen.sha(deltas[:m].cpu().numpy()), fx[f"march/{cfg}/sha_deltas"])
    else:
        # rays the budget dropped write nothing (raymarching.cu:422); rows of kept rays are exact
        rr = fx[f"march/{cfg}/rays"]
        kept = rr[(rr[:, 2] > 0) & (rr[:, 1] + rr[:, 2] <= M)]
        end = int((kept[:, 1] + kept[:, 2]).max())
        assert end <= M and not xyzs[end:].any()


def test_composite_rays_train(cuda, fx, native):
    rm, _ = native
    xyzs, dirs, deltas, rays, counter, M = _march(rm, fx, "perturb", cuda)
    Mc, N = int(fx["comp/M"]), rays.shape[0]
    g = np.random.default_rng(int(fx["comp/seed"]))
    sig = _t(np.exp(g.standard_normal(Mc) * 2.0).astype(np.float32), cuda)
    rgb = _t(g.random((Mc, 3)).astype(np.float32), cuda)
    rr = _t(fx["comp/rays"], cuda)
    dl = deltas[:Mc].contiguous()
    ws, dep, img = torch.empty(N, device=cuda), torch.empty(N, device=cuda), torch.empty(N, 3, device=cuda)
    rm.composite_rays_train_forward(sig, rgb, dl, rr, Mc, N, 1e-4, ws, dep, img)
    np.testing.assert_allclose(ws.cpu().numpy(), fx["comp/ws"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(img.cpu().numpy(), fx["comp/image"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(dep.cpu().numpy(), fx["comp/depth"], rtol=5e-5, atol=1e-5)
    assert float(ws[9]) == 0 and float(img[9].abs().sum()) == 0                      # the overflowing ray
    gws = _t(g.standard_normal(N).astype(np.float32), cuda)
    gimg = _t(g.standard_normal((N, 3)).astype(np.float32), cuda)
    gs, gc = torch.zeros(Mc, device=cuda), torch.zeros(Mc, 3, device=cuda)
    rm.composite_rays_train_backward(gws, gimg, sig, rgb, dl, rr, ws, img, Mc, N, 1e-4, gs, gc)
    rows = fx["march/perturb/sub_rows"]
    np.testing.assert_allclose(gc.cpu().numpy()[rows], fx["comp/sub_gc"], rtol=2e-5, atol=2e-6)
    scale = np.abs(fx["comp/sub_gs"]).max()
    np.testing.assert_allclose(gs.cpu().numpy()[rows], fx["comp/sub_gs"], rtol=1e-3, atol=2e-5 * scale)
    assert abs(float(gs.double().abs().sum()) - float(fx["comp/sum_abs_gs"])) < 1e-4 * float(fx["comp/sum_abs_gs"])
    np.testing.assert_allclose(gc.double().sum(0).cpu().numpy(), fx["comp/sum_gc"], rtol=1e-4, atol=1e-3)


def test_inference_loop(cuda, fx, native):
    """run_cuda's eval loop (renderer.py:324-374) over march_rays / composite_rays with the fixture's analytic field;
    surviving ray ids per iteration exact (hash), outputs to fp32 rounding."""
    rm, _ = native
    o, d = _t(fx["infer/o"], cuda), _t(fx["infer/d"], cuda)
    nears, fars = _t(fx["infer/nears"], cuda), _t(fx["infer/fars"], cuda)
    bits = _t(fx["bitfield"], cuda)
    N = o.shape[0]
    ws, dep, img = torch.zeros(N, device=cuda), torch.zeros(N, device=cuda), torch.zeros(N, 3, device=cuda)
    alive = torch.arange(N, dtype=torch.int32, device=cuda)
    rt = nears.clone()
    step, hist, h = 0, [], hashlib.sha256()
    while step < gen.MAX_STEPS:
        n_alive = alive.shape[0]
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        M = n_alive * n_step
        M += 128 - (M % 128)                                                        # raymarching.py:329-331, align 128
        xyzs, dirs = torch.zeros(M, 3, device=cuda), torch.zeros(M, 3, device=cuda)
        deltas = torch.zeros(M, 2, device=cuda)
        rm.march_rays(n_alive, n_step, alive, rt, o, d, gen.BOUND, 0.0, gen.MAX_STEPS, gen.CAS, gen.HG, bits, nears, fars,
                      xyzs, dirs, deltas, torch.zeros(n_alive, device=cuda))
        s, c = gen.analytic_field(xyzs.cpu().numpy(), dirs.cpu().numpy())
        rm.composite_rays(n_alive, n_step, 1e-2, alive, rt, _t(s, cuda), _t(c, cuda), deltas, ws, dep, img)
        alive = alive[alive >= 0]
        hist.append(alive.shape[0])
        h.update(alive.cpu().numpy().tobytes())
        step += n_step
    assert hist == fx["infer/n_alive"].tolist()
    assert np.array_equal(np.frombuffer(h.digest(), np.uint8), fx["infer/alive_sha"])
    np.testing.assert_allclose(img.cpu().numpy(), fx["infer/image"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ws.cpu().numpy(), fx["infer/ws"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dep.cpu().numpy(), fx["infer/depth"], rtol=1e-4, atol=1e-5)
