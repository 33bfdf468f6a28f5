"""Live / deferred split of the coefficient optimiser pass (TrainStep defer_adam, csrc/adam.hip):
the replayed updates must equal the per-step pass bit for bit, and nothing the windowed step reads or writes may lie
outside the live rectangles."""
import copy
import ctypes as C_

import numpy as np
import pytest
import torch

from trinerflet_amd import synthetic

pytestmark = pytest.mark.gpu


def _rect(*v):
    return (C_.c_int32 * 8)(*v)


@pytest.mark.parametrize("l1", [0.0, 3e-9])
def test_live_plus_catchup_equals_per_step_pass_bit_for_bit(cuda, l1):
    """K steps of tnl_adam_l1_step_rect over a whole level against K x (record + live rectangle) + one replay:
    p, m, v identical to the bit, including a GradScaler-skipped step; the L1 sums agree (summation order differs)."""
    import trinerflet_amd._lib as L
    lib = L.lib()
    S, bands, n, spp, K = 6, 3, 64, 2, 7
    g = torch.Generator(device="cpu").manual_seed(4)
    numel = S * bands * n * n
    p0 = torch.randn(numel, generator=g).to(cuda) * 0.1
    p0[::17] = 0.0                                                   # sign(0) = 0 coefficients
    m0 = torch.randn(numel, generator=g).to(cuda) * 1e-3
    v0 = torch.rand(numel, generator=g).to(cuda) * 1e-6
    live = [16, 0, 32, 8, 16, 0, 32, 40]                             # ox[3], oy[3], w, h
    grect = [20, 4, 36, 8, 24, 8, 24, 24]                            # inside live for every plane
    grads = [torch.randn(numel, generator=g).to(cuda) * 64.0 for _ in range(K)]
    found = [0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0]                      # step 2 is skipped by the scaler
    inv_scale = torch.full((1,), 1.0 / 64.0, device=cuda)
    b1, b2, eps, lr = 0.9, 0.99, 1e-15, 1e-2

    def run(deferred):
        p, m, v = p0.clone(), m0.clone(), v0.clone()
        opt = torch.zeros(1, device=cuda)
        ring = torch.zeros(64, device=cuda)
        sums = torch.zeros(16, device=cuda)
        per_step = []
        for k in range(K):
            fi = torch.full((1,), found[k], device=cuda)
            ab = torch.zeros(1, device=cuda)
            lr_k = lr * (0.97 ** k)
            if deferred:
                L.check(lib.tnl_adam_record_step(L.ptr(ring), L.i32(k), L.f32(lr_k), L.ptr(opt), L.f32(b1), L.f32(b2),
                                                 L.ptr(fi), L.stream()), "record")
                L.check(lib.tnl_adam_l1_step_live(
                    L.ptr(p), L.ptr(grads[k]), L.ptr(m), L.ptr(v), L.u32(S), L.u32(spp), L.u32(0), L.u32(1),
                    (C_.c_uint64 * 1)(0), (C_.c_uint32 * 1)(n), (C_.c_uint32 * 1)(bands), _rect(*live), _rect(*grect),
                    (C_.c_float * 1)(l1), L.f32(lr_k), L.ptr(opt), L.ptr(ring[4 * k:]) if k % 2 else None, L.f32(b1),
                    L.f32(b2), L.f32(eps), L.f32(1.0), L.ptr(inv_scale), L.ptr(fi), L.ptr(ab), L.stream()), "live")
            else:
                L.check(lib.tnl_adam_l1_step_rect(L.ptr(p), L.ptr(grads[k]), L.ptr(m), L.ptr(v), L.u32(S), L.u32(bands),
                                                  L.u32(n), L.u32(spp), L.u32(0), _rect(*grect), L.f32(lr_k), L.ptr(opt),
                                                  L.f32(b1), L.f32(b2), L.f32(eps), L.f32(1.0), L.ptr(inv_scale),
                                                  L.f32(l1), L.ptr(fi), L.ptr(ab), L.stream()), "rect")
            opt += 1.0 - found[k]                                    # the step epilogue's counter
            per_step.append(ab)
        if deferred:
            L.check(lib.tnl_adam_l1_catchup(L.ptr(p), L.ptr(m), L.ptr(v), L.u32(S), L.u32(bands), L.u32(n), L.u32(spp),
                                            L.u32(0), _rect(*live), L.ptr(ring), L.i32(K), L.f32(b1), L.f32(b2),
                                            L.f32(eps), L.f32(l1), L.ptr(sums), L.stream()), "catchup")
            per_step = [a + sums[k] for k, a in enumerate(per_step)]
        torch.cuda.synchronize()
        return p, m, v, torch.cat(per_step)

    ref, dfr = run(False), run(True)
    inside = torch.zeros(S, bands, n, n, dtype=torch.bool, device=cuda)
    for sl in range(S):
        pl = sl // spp
        inside[sl, :, live[3 + pl]:live[3 + pl] + live[7], live[pl]:live[pl] + live[6]] = True
    for a, b, name in zip(ref[:3], dfr[:3], "pmv"):
        ne = (a != b).view(S, bands, n, n)
        assert torch.equal(a, b), (name, int(ne.sum()), "inside the live rectangle:", int((ne & inside).sum()),
                                   float((a - b).abs().max()))
    assert not torch.equal(ref[0], p0)
    np.testing.assert_allclose(dfr[3].cpu().numpy(), ref[3].cpu().numpy(), rtol=2e-5)
    # something really was deferred: the live pass alone leaves the outside untouched
    view = ref[0].view(S, bands, n, n)
    assert not torch.equal(view[0, 0, 50:, 50:], p0.view(S, bands, n, n)[0, 0, 50:, 50:])


def _model(dev, C=16, R=512, scale=8, H=64, bound=1.0):
    from trinerflet_amd.nerf.network import NeRFNetwork
    m = NeRFNetwork(encoding="triplane_wavelet", bound=bound, cuda_ray=True, density_thresh=10, hidden_dim=H,
                    hidden_dim_color=H, triplane_channels=C, triplane_resolution=R, triplane_wavelet_levels=scale,
                    wavelet_type="bior6.8").to(dev)
    synthetic.init_field_parameters(m, seed=3)
    return m


def _setup(cuda, N=2048, bound=1.0, radius=0.3, **model_kw):
    o, d = synthetic.training_rays(N, n_cams=4, seed=7)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    gt = t(synthetic.target_colors(d))
    noise = t(np.random.default_rng(0).random(N).astype(np.float32))
    base = _model(cuda, bound=bound, **model_kw)
    bf = t(synthetic.sphere_bitfield(128, 1, bound, radius, 0.0))
    base.density_bitfield.copy_(bf)
    return t(o), t(d), gt, noise, base, bf


def test_training_with_deferred_pass_equals_per_step_pass(cuda):
    """Eleven steps (grid refreshes at 0, 4, 8) with and without the deferral.  Outside the live rectangles a
    coefficient's trajectory depends on nothing but its own p, m, v and the steps' scalars: bit-identical between the
    two runs.  Inside, the runs differ by the tile reduction's run-to-run summation order only."""
    from trinerflet_amd.train import TrainStep
    o, d, gt, noise, base, bf = _setup(cuda)
    res = []
    for defer in (False, True, False):   # the second per-step run is the yardstick of run-to-run noise
        m = copy.deepcopy(base)
        ts = TrainStep(m, update_extra_interval=4, use_roi=True, defer_adam=defer)
        ts.post_refresh = lambda m=m: m.density_bitfield.copy_(bf)
        m.mean_count = 0
        losses, lives = [], None
        for it in range(11):
            losses.append(ts.step(o, d, gt, noises=noise).clone())
            if defer:   # a refresh step (whole planes in, windowed gradient out) opens the next period itself
                assert ts._pending == it % 4 + 1 and any(lv is not None for lv in ts._live)
                lives = [None if lv is None else list(lv) for lv in ts._live]
        total = torch.stack(losses).sum() + ts.pop_deferred_reg()
        assert ts._pending == 0
        res.append((float(total), [float(l) for l in losses], [p.detach().clone() for p in ts.coef.params],
                    [x.clone() for x in (ts.coef.m, ts.coef.v)], lives, ts))
    ts = res[1][5]
    assert ts.defer_adam and ts.deferred_steps == 11 and ts.deferred_flushes == 3 and not res[0][5].defer_adam
    np.testing.assert_allclose(res[0][0], res[2][0], rtol=2e-4)
    lives = res[1][4]
    assert lives is not None and sum(lv is not None for lv in lives) >= 1
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=2e-4)          # sum of losses incl. the deferred L1 share
    for lvl, (a, b) in enumerate(zip(res[0][2], res[1][2])):
        n = a.shape[-1]
        if lives[lvl] is None:
            continue
        outside = torch.ones(3, 1, 1, n, n, dtype=torch.bool, device=cuda)
        lv = lives[lvl]
        for p in range(3):
            outside[p, :, :, lv[3 + p]:lv[3 + p] + lv[7], lv[p]:lv[p] + lv[6]] = False
        outside = outside.expand_as(a)
        assert float(outside.float().mean()) > 0.2
        assert torch.equal(a[outside], b[outside]), lvl                   # replayed == stepped, to the bit
        # inside: Adam (eps = 1e-15) turns a gradient at the tile reduction's noise level into +-lr steps; the
        # same statistic between two per-step runs is the yardstick
        c = res[2][2][lvl]
        far = lambda x, y: int(((x - y).abs() > 2e-3 + 1e-3 * y.abs()).sum())
        # (the count itself scatters from run to run -- 92 and 400 of 9.4 M were seen for the same pair of
        #  configurations -- so the bound is generous; the exact statement is the one about the outside above)
        assert far(b, a) <= 10 * far(c, a) + int(2e-4 * a.numel()), (lvl, far(b, a), far(c, a))
        assert float((a - b).abs().max()) < 2 * 11 * 1e-2
    for a, b in zip(res[0][3], res[1][3]):                                # moments: same statement over the flat arrays
        assert float((a - b).abs().max()) < 1.0 and torch.isfinite(b).all()


@pytest.mark.parametrize("geom", [dict(C=16, R=512, scale=8), dict(C=16, R=1024, scale=16)])
def test_windowed_step_reads_nothing_outside_the_live_rectangles(cuda, geom):
    """NaN in every coefficient outside the live rectangles: the windowed plane rebuild must produce the same bits
    (R = 1024: the finest level runs the column-walk kernels)."""
    from trinerflet_amd.train import TrainStep
    o, d, gt, noise, base, bf = _setup(cuda, **geom)
    m = copy.deepcopy(base)
    ts = TrainStep(m, update_extra_interval=16, use_roi=True, defer_adam=True)
    ts.post_refresh = lambda m=m: m.density_bitfield.copy_(bf)
    m.mean_count = 0
    ts.step(o, d, gt, noises=noise)
    ts.step(o, d, gt, noises=noise)
    assert ts._pending == 2
    lives = [None if lv is None else list(lv) for lv in ts._live]
    assert sum(lv is not None for lv in lives) >= 1
    ts.flush_deferred()
    ts.rebuild_planes(roi=True)
    want = ts._tm_full.clone()
    with torch.no_grad():
        for lvl, lv in enumerate(lives):
            if lv is None:
                continue
            cf = ts.coef.params[lvl]
            keep = cf.detach().clone()
            cf.fill_(float("nan"))
            for p in range(3):
                ys, xs = slice(lv[3 + p], lv[3 + p] + lv[7]), slice(lv[p], lv[p] + lv[6])
                cf[p, :, :, ys, xs] = keep[p, :, :, ys, xs]
    ts.rebuild_planes(roi=True)
    assert torch.isfinite(ts._tm_full.float()).all()
    assert torch.equal(ts._tm_full, want)
    # and the windowed adjoint's rectangles lie inside the live ones
    for lvl, lv in enumerate(lives):
        r = ts._rects[lvl]
        if lv is None:
            continue
        for p in range(3):
            assert lv[p] <= r[p] and r[p] + r[6] <= lv[p] + lv[6] and lv[3 + p] <= r[3 + p] and \
                r[3 + p] + r[7] <= lv[3 + p] + lv[7]
