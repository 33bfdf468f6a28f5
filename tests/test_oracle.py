"""CPU tests: the oracle against every golden vector we hold, and against domain properties.

Goldens (tests/golden/, generating scripts alongside):
  idwt_pywt.npz            real PyWavelets pywt.idwt2(mode='zero') (make_golden_pywt.py)
  triplane_reference.npz   the REFERENCE TriPlaneVolume (get_planes, forward + autograd VJP) and trunc_exp,
                           run in-container from /root/reference (make_golden_reference.py)
"""
import os

import numpy as np
import pytest
import torch

from oracle import cref, field as ofield
from trinerflet_amd import synthetic


@pytest.fixture(scope="module")
def gref(golden_dir):
    return np.load(os.path.join(golden_dir, "triplane_reference.npz"))


@pytest.fixture(scope="module")
def gpywt(golden_dir):
    return np.load(os.path.join(golden_dir, "idwt_pywt.npz"))


@pytest.mark.parametrize("wave", cref.WAVELETS)
def test_taps_and_idwt_vs_pywt(gpywt, wave):
    _, L, lo, hi = cref.wavelet_taps(wave)
    assert np.array_equal(lo, gpywt[f"{wave}/rec_lo"]) and np.array_equal(hi, gpywt[f"{wave}/rec_hi"])
    x = gpywt[f"{wave}/ll"]
    for lvl in range(2):
        x = cref.idwt_level(x, gpywt[f"{wave}/yh{lvl}"], wave, f64=True, taps_f32=False)
    assert np.abs(x - gpywt[f"{wave}/planes"]).max() < 1e-13


@pytest.mark.parametrize("wave", cref.WAVELETS)
def test_build_planes_vs_reference(gref, wave):
    ll, c0, c1 = gref[f"idwt/{wave}/ll"], gref[f"idwt/{wave}/coef0"], gref[f"idwt/{wave}/coef1"]
    x = ll.reshape(6, 8, 8)
    for c in (c0, c1):
        n = x.shape[-1]
        x = cref.idwt_level(x, c.reshape(6, 3, n, n), wave, f64=True, taps_f32=False)
    assert np.abs(x.reshape(3, 2, 32, 32) - gref[f"idwt/{wave}/planes"]).max() < 1e-13
    # float32 path (what the GPU is compared with) stays within fp32 rounding of it
    p32 = cref.build_planes(ll, [c0, c1], wave)
    assert np.abs(p32 - gref[f"idwt/{wave}/planes"]).max() < 2e-6 * np.abs(gref[f"idwt/{wave}/planes"]).max()
    # the torch conv_transpose2d restatement used by the CPU baseline computes the same thing
    pt = ofield.build_planes_torch(torch.from_numpy(ll), [torch.from_numpy(c0), torch.from_numpy(c1)], wave)
    assert np.abs(pt.numpy() - gref[f"idwt/{wave}/planes"]).max() < 1e-6 * max(1.0, np.abs(pt.numpy()).max())


@pytest.mark.parametrize("wave", cref.WAVELETS)
def test_adjoint_identity(wave):
    rng = np.random.default_rng(0)
    S, n = 3, 12
    x, yh = rng.standard_normal((S, n, n)), rng.standard_normal((S, 3, n, n))
    y = rng.standard_normal((S, 2 * n, 2 * n))
    Ax = cref.idwt_level(x, yh, wave, f64=True)
    dx, dyh = cref.idwt_level_adj(y, wave, f64=True)
    assert abs((Ax * y).sum() - ((x * dx).sum() + (yh * dyh).sum())) < 1e-9
    # and torch autograd of the conv_transpose2d restatement agrees with the closed-form adjoint
    xt = torch.from_numpy(x).view(1, S, n, n).requires_grad_(True)
    yt = torch.from_numpy(yh).view(1, S, 3, n, n).requires_grad_(True)
    out = ofield.idwt_level_torch(xt, yt, wave)
    out.backward(torch.from_numpy(y).view(1, S, 2 * n, 2 * n))
    assert np.abs(xt.grad.numpy().reshape(S, n, n) - dx).max() < 1e-5
    assert np.abs(yt.grad.numpy().reshape(S, 3, n, n) - dyh).max() < 1e-5


def test_sample_vs_reference(gref):
    f = cref.triplane_sample(gref["sample/planes"], gref["sample/xyz"], float(gref["sample/bound"]))
    assert np.abs(f - gref["sample/feats"]).max() < 5e-7
    d = cref.triplane_sample_bwd(gref["sample/cot"], gref["sample/xyz"], float(gref["sample/bound"]), 4, 32)
    assert np.abs(d - gref["sample/dplanes"]).max() < 2e-6
    ft = ofield.triplane_features(torch.from_numpy(gref["sample/planes"]), torch.from_numpy(gref["sample/xyz"]),
                                  float(gref["sample/bound"]))
    assert np.abs(ft.numpy() - gref["sample/feats"]).max() < 5e-7


def test_trunc_exp_vs_reference(gref):
    x = torch.from_numpy(gref["trunc_exp/x"]).requires_grad_(True)
    y = ofield._TruncExp.apply(x)
    (gx,) = torch.autograd.grad(y, x, torch.from_numpy(gref["trunc_exp/g"]))
    assert np.allclose(y.detach().numpy(), gref["trunc_exp/y"], rtol=1e-6)
    assert np.allclose(gx.numpy(), gref["trunc_exp/gx"], rtol=1e-6)


def test_sh4_orthonormal():
    """The 16 real SH basis functions are orthonormal on the sphere (property; the CUDA source cannot run)."""
    rng = np.random.default_rng(0)
    d = rng.standard_normal((400000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    Y = cref.sh4(d).astype(np.float64)
    G = 4 * np.pi * (Y.T @ Y) / d.shape[0]
    assert np.abs(G - np.eye(16)).max() < 2e-2
    assert np.allclose(ofield.sh4(torch.from_numpy(d[:100])).numpy(), cref.sh4(d[:100]), atol=1e-6)


def test_morton_and_packbits():
    ax = np.arange(128, dtype=np.int32)
    coords = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    idx = cref.morton3D(coords)
    assert np.array_equal(np.sort(idx), np.arange(128 ** 3))
    assert np.array_equal(cref.morton3D_invert(idx), coords)
    g = np.random.default_rng(0).standard_normal(8 * 1000).astype(np.float32)
    bits = cref.packbits(g, 0.25)
    assert np.array_equal(np.unpackbits(bits, bitorder="little").astype(bool), g > 0.25)
    # the synthetic analytic bitfield uses the same bit order: centre cell of cascade 0 is occupied, corner is not
    bf = synthetic.sphere_bitfield(128, 2, 1.5, 0.8, 0.0)
    c = int(cref.morton3D(np.array([[64, 64, 64]], np.int32))[0])
    k = int(cref.morton3D(np.array([[0, 0, 0]], np.int32))[0])
    assert bf[c // 8] & (1 << (c % 8)) and not bf[k // 8] & (1 << (k % 8))


@pytest.fixture(scope="module")
def marched():
    o, d = synthetic.training_rays(600, n_cams=4, seed=2)
    o[:3] += 50.0  # rays that miss the box
    aabb = np.array([-1.5] * 3 + [1.5] * 3, np.float32)
    nears, fars = cref.near_far_from_aabb(o, d, aabb, 0.2)
    bf = synthetic.sphere_bitfield(128, 2, 1.5, 0.8, 0.6)
    noise = np.random.default_rng(1).random(600).astype(np.float32)
    return o, d, nears, fars, bf, noise, cref.march_rays_train(o, d, 1.5, bf, 2, 128, nears, fars, noise, 600 * 1024)


def test_march_properties(marched):
    o, d, nears, fars, bf, noise, (xyz, dirs, deltas, rays, counter) = marched
    assert np.all(nears[:3] == np.finfo(np.float32).max) and np.all(rays[:3, 2] == 0)
    assert counter[1] == 600 and counter[0] == rays[:, 2].sum()
    assert np.array_equal(rays[:, 0], np.arange(600))
    assert np.array_equal(rays[:, 1], np.concatenate([[0], np.cumsum(rays[:, 2])[:-1]]))  # a valid packing
    total = int(counter[0])
    r = np.linalg.norm(xyz[:total], axis=-1)
    assert r.min() > 0.55 and r.max() < 0.85          # samples lie in the occupied shell (cell-size slack)
    assert np.allclose(deltas[:total, 0], np.float32(2 * np.sqrt(3) / 1024))  # dt_gamma = 0 -> dt_min
    assert np.all(xyz[total:] == 0)                    # untouched rows stay zero (they feed the MLP)
    assert (rays[:, 2] > 0).sum() > 50
    n = int(np.argmax(rays[:, 2]))                      # the ray with the most samples
    off, cnt = rays[n, 1], rays[n, 2]
    assert cnt > 0 and np.allclose(dirs[off:off + cnt], d[n])
    t = np.einsum("ij,j->i", xyz[off:off + cnt] - o[n], d[n])
    assert np.all(np.diff(t) > 0) and t[0] >= nears[n] and t[-1] < fars[n]
    # overflow rule: a smaller budget drops whole rays, never truncates one (raymarching.cu:421-422)
    M = total // 2
    x2, _, _, rays2, _ = cref.march_rays_train(o, d, 1.5, bf, 2, 128, nears, fars, noise, M)
    kept = rays2[:, 1] + rays2[:, 2] <= M
    last = int((rays2[kept, 1] + rays2[kept, 2]).max())
    assert np.array_equal(x2[:last], xyz[:last]) and np.all(x2[last:] == 0)


def test_composite_vs_torch_and_cumprod(marched):
    o, d, nears, fars, bf, noise, (xyz, dirs, deltas, rays, counter) = marched
    total = int(counter[0])
    rng = np.random.default_rng(3)
    sig = np.exp(rng.standard_normal(total)).astype(np.float32) * 3
    rgb = rng.random((total, 3)).astype(np.float32)
    dl = deltas[:total]
    ws, dep, img = cref.composite_rays_train_forward(sig, rgb, dl, rays, 1e-4)
    # (a) sequential torch restatement, differentiable: forward and the analytic backward of the kernel
    sub = rays[100:140]
    st = torch.from_numpy(sig).double().requires_grad_(True)
    ct = torch.from_numpy(rgb).double().requires_grad_(True)
    sub_local = sub.copy()
    sub_local[:, 0] = np.arange(sub.shape[0])
    w2, d2, i2 = ofield.composite_train_torch(st, ct, torch.from_numpy(dl).double(), sub_local, 1e-4)
    assert np.allclose(w2.detach().numpy(), ws[sub[:, 0]], atol=1e-5)
    assert np.allclose(i2.detach().numpy(), img[sub[:, 0]], atol=1e-5)
    assert np.allclose(d2.detach().numpy(), dep[sub[:, 0]], atol=1e-4)
    gw = rng.standard_normal(sub.shape[0])
    gi = rng.standard_normal((sub.shape[0], 3))
    ((w2 * torch.from_numpy(gw)).sum() + (i2 * torch.from_numpy(gi)).sum()).backward()
    gws = np.zeros(600, np.float32)
    gim = np.zeros((600, 3), np.float32)
    gws[sub[:, 0]], gim[sub[:, 0]] = gw, gi
    gs, gc = cref.composite_rays_train_backward(gws, gim, sig, rgb, dl, rays, ws, img, 1e-4)
    lo, hi = sub[0, 1], sub[-1, 1] + sub[-1, 2]
    assert np.allclose(gc[lo:hi], ct.grad.numpy()[lo:hi], atol=1e-5)
    assert np.allclose(gs[lo:hi], st.grad.numpy()[lo:hi], atol=2e-4)
    # (b) with T_thresh = 0 the recurrence equals the reference's pure-torch compositing
    #     weights = alpha * cumprod([1, 1 - alpha + 1e-15])[:-1]   (renderer.py:206-210)
    ws0, _, img0 = cref.composite_rays_train_forward(sig, rgb, dl, rays, 0.0)
    for n in (120, 300, 555):
        off, cnt = rays[n, 1], rays[n, 2]
        a = 1 - np.exp(-sig[off:off + cnt].astype(np.float64) * dl[off:off + cnt, 0])
        w = a * np.cumprod(np.concatenate([[1.0], 1 - a + 1e-15]))[:-1]
        assert abs(w.sum() - ws0[n]) < 1e-5 and np.abs((w[:, None] * rgb[off:off + cnt]).sum(0) - img0[n]).max() < 1e-5


def test_inference_loop_equals_train_march(marched):
    """The alive-ray loop (renderer.py:324-374) visits exactly the samples of the training march and, with a
    threshold that never triggers, composites to the same image."""
    o, d, nears, fars, bf, noise, _ = marched
    N = 64
    o, d, nears, fars = o[100:100 + N], d[100:100 + N], nears[100:100 + N], fars[100:100 + N]
    zero = np.zeros(N, np.float32)
    xyz, _, deltas, rays, counter = cref.march_rays_train(o, d, 1.5, bf, 2, 128, nears, fars, zero, N * 1024)
    total = int(counter[0])

    def field(x):
        return (2.0 * np.exp(-(x ** 2).sum(-1))).astype(np.float32), (0.5 + 0.5 * np.cos(2 * x)).astype(np.float32)
    s, c = field(xyz[:total])
    ws_t, dep_t, img_t = cref.composite_rays_train_forward(s, c, deltas[:total], rays, 0.0)
    ws, dep, img = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    alive, rt, step, visited = np.arange(N, dtype=np.int32), nears.copy(), 0, 0
    while step < 1024 and alive.shape[0] > 0:
        n_alive = alive.shape[0]
        n_step = max(min(N // n_alive, 8), 1)
        x, _, dl = cref.march_rays(n_alive, n_step, alive, rt, o, d, 1.5, bf, 2, 128, nears, fars,
                                   np.zeros(n_alive, np.float32))
        visited += int((dl[:, 0] > 0).sum())
        s, c = field(x)
        cref.composite_rays(n_alive, n_step, alive, rt, s, c, dl, ws, dep, img, -1.0)
        alive = alive[alive >= 0]
        step += n_step
    assert visited == total
    assert np.allclose(img, img_t, atol=2e-5) and np.allclose(ws, ws_t, atol=2e-5)
    # the training kernel accumulates depth from t = 0 (raymarching.cu:535,545), the inference kernel from the
    # absolute ray time (:845,872): they differ by ws * near, as in the reference
    assert np.allclose(dep, dep_t + ws_t * nears, atol=2e-4)


def test_lr_schedule_and_reg():
    # decay_function (utils.py:55-62): warm-up ramp then 0.1^(progress^2.5)
    assert abs(ofield.lr_factor(0, 1000, 0) - 1.0) < 1e-12
    assert abs(ofield.lr_factor(1000, 1000, 0) - 0.1) < 1e-12
    assert abs(ofield.lr_factor(0, 1000, 100) - 0.1 * 1e-3) < 1e-12
    assert abs(ofield.lr_factor(99, 1000, 100) - (1e-4 + 99 * (1 - 1e-3) / 99)) < 1e-12
    from trinerflet_amd.train import lr_factor
    for it in (0, 5, 99, 100, 600, 5000):
        assert lr_factor(it, 1000, 100) == ofield.lr_factor(it, 1000, 100)
    c = [torch.randn(3, 2, 3, 8, 8), torch.randn(3, 2, 3, 16, 16)]
    total = sum(x.numel() for x in c)
    assert abs(float(ofield.wavelet_reg(c, 0.4)) - 0.4 / (2 * total) * float(sum(x.abs().sum() for x in c))) < 1e-6
