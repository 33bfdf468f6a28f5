"""GPU parity of the raymarching / shencoder kernels against the CPU oracle (through the C ABI)."""
import numpy as np
import pytest
import torch

from oracle import cref
from trinerflet_amd import synthetic as scene

pytestmark = pytest.mark.gpu

BOUND, CAS, HG = 1.5, 2, 128


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(scope="module")
def rays(cuda):
    o, d = scene.training_rays(4096, n_cams=8, seed=3)
    # a few rays that miss the box and axis-parallel rays (rd = inf)
    d[:4] = np.array([[0, 1, 0], [1, 0, 0], [0, 0, -1], [0.6, 0.8, 0.0]], np.float32)
    o[4:8] += 10.0
    aabb = np.array([-BOUND] * 3 + [BOUND] * 3, np.float32)
    nears, fars = cref.near_far_from_aabb(o, d, aabb, 0.2)
    return o, d, aabb, nears, fars


def test_near_far(cuda, rays):
    from trinerflet_amd import raymarching
    o, d, aabb, nears, fars = rays
    n, f = raymarching.near_far_from_aabb(_t(o, cuda), _t(d, cuda), _t(aabb, cuda), 0.2)
    assert np.array_equal(n.cpu().numpy(), nears) and np.array_equal(f.cpu().numpy(), fars)


def test_morton_roundtrip_and_packbits(cuda):
    from trinerflet_amd import raymarching
    g = np.random.default_rng(0)
    coords = g.integers(0, 128, (100000, 3)).astype(np.int32)
    idx = raymarching.morton3D(_t(coords, cuda))
    assert np.array_equal(idx.cpu().numpy(), cref.morton3D(coords))
    back = raymarching.morton3D_invert(idx)
    assert np.array_equal(back.cpu().numpy(), coords)
    # all 128^3 codes: a permutation (checksum) -- SURVEY 8(c) F-GRID
    ax = torch.arange(128, dtype=torch.int32, device=cuda)
    full = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    codes = raymarching.morton3D(full).long()
    assert torch.equal(torch.sort(codes).values, torch.arange(128 ** 3, device=cuda))
    for n in (128 ** 3 * 2, 8 * 37):  # vector path and ragged tail path
        grid = g.standard_normal(n).astype(np.float32).reshape(1, -1)
        bf = raymarching.packbits(_t(grid, cuda), 0.1)
        assert np.array_equal(bf.cpu().numpy(), cref.packbits(grid, 0.1))


@pytest.mark.parametrize("form", [0, 1])
@pytest.mark.parametrize("shell", [False, True])
@pytest.mark.parametrize("perturb", [False, True])
def test_march_train_exact(cuda, rays, shell, perturb, form):
    """form: tnl_march_count_form -- 0 the wavefront-per-ray count pass (64 consecutive chain points per wave), 1 one ray
    per lane; both against the oracle's serial march, bit for bit."""
    from trinerflet_amd import raymarching
    with raymarching.count_form(form):
        _march_train_exact(cuda, rays, shell, perturb)
    from trinerflet_amd import _lib as L
    assert L.lib().tnl_march_count_form(L.i32(-1)) == 0           # restored (an argument other than 0 / 1 only queries)


def _march_train_exact(cuda, rays, shell, perturb):
    from trinerflet_amd import raymarching
    o, d, aabb, nears, fars = rays
    bf = scene.sphere_bitfield(HG, CAS, BOUND, 0.8, 0.7 if shell else 0.0)
    N = o.shape[0]
    noises = np.random.default_rng(5).random(N).astype(np.float32) if perturb else np.zeros(N, np.float32)
    M_full = N * 1024
    xr, dr, lr, rr, cr = cref.march_rays_train(o, d, BOUND, bf, CAS, HG, nears, fars, noises, M_full)
    total = int(cr[0])
    assert total > 0
    for M in (M_full, total // 2):  # second case exercises the overflow drop rule (raymarching.cu:422)
        if M != M_full:
            xr, dr, lr, rr, cr = cref.march_rays_train(o, d, BOUND, bf, CAS, HG, nears, fars, noises, M)
        counter = torch.zeros(2, dtype=torch.int32, device=cuda)
        # mean_count = M - align  -> the wrapper rounds up past the next multiple (raymarching.py:200-203)
        xyzs, dirs, deltas, rays_t = raymarching.march_rays_train(
            _t(o, cuda), _t(d, cuda), BOUND, _t(bf, cuda), CAS, HG, _t(nears, cuda), _t(fars, cuda), counter,
            -1 if M == M_full else M, perturb, -1, M == M_full, 0, 1024, _t(noises, cuda))
        assert np.array_equal(counter.cpu().numpy(), cr)            # bit-exact counts
        assert np.array_equal(rays_t.cpu().numpy(), rr)             # ids, offsets, num_steps
        m = xyzs.shape[0]
        assert np.array_equal(xyzs.cpu().numpy(), xr[:m])           # bit-exact sample positions
        assert np.array_equal(dirs.cpu().numpy(), dr[:m])
        assert np.array_equal(deltas.cpu().numpy(), lr[:m])


@pytest.mark.parametrize("case", ["one_cascade", "bound2", "cap64", "near_tiny", "cap7_dense", "random_cells", "few_partial"])
def test_march_wavefront_form_edge_cases(cuda, case):
    """The wavefront-per-ray count pass where its closed-form chain does not apply or its bookkeeping is stressed: one
    cascade, bound 2 (long rays: t crosses 2 and 4), a sample cap of 64 / 7 per ray on a fully occupied grid (the cap ends a run inside a chunk),
    min_near 0.01 (t starts in binades where 63 steps are no longer exact: the serial chunk form)."""
    from trinerflet_amd import raymarching
    # (round 6, the per-lane form's occupancy table in LDS: "random_cells" = every 64-cell word partial, 65 536 of them, more
    #  than the table holds: the words come from memory; "few_partial" = a thin shell, every partial word in LDS; the full
    #  grids above = only all-ones words)
    bound, cas, max_steps, min_near, shell = {"one_cascade": (1.0, 1, 1024, 0.2, (0.8, 0.3)), "bound2": (2.0, 2, 1024, 0.2, (0.9, 0.0)),
                                              "cap64": (1.5, 2, 64, 0.2, None), "near_tiny": (1.5, 2, 512, 0.01, (1.4, 0.0)),
                                              "cap7_dense": (1.5, 2, 7, 0.2, None), "random_cells": (1.5, 2, 256, 0.2, "random"),
                                              "few_partial": (1.5, 2, 1024, 0.2, (0.5, 0.45))}[case]
    o, d = scene.training_rays(3000, n_cams=6, seed=9)
    if case == "near_tiny":
        o *= 0.3            # cameras inside the box: the rays start at t = min_near
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = cref.near_far_from_aabb(o, d, aabb, min_near)
    if shell is None:
        bf = np.full(cas * HG ** 3 // 8, 255, np.uint8)
    elif shell == "random":
        rb = np.random.default_rng(11)
        bf = (rb.integers(0, 256, cas * HG ** 3 // 8) & rb.integers(0, 256, cas * HG ** 3 // 8)).astype(np.uint8)   # 25 % of the cells
    else:
        bf = scene.sphere_bitfield(HG, cas, bound, *shell)
    noises = np.random.default_rng(2).random(o.shape[0]).astype(np.float32)
    M = o.shape[0] * max_steps
    xr, dr, lr, rr, cr = cref.march_rays_train(o, d, bound, bf, cas, HG, nears, fars, noises, M, max_steps=max_steps)
    assert int(cr[0]) > 1000
    if case.startswith("cap"):
        assert (rr[:, 2] == max_steps).mean() > 0.05          # the cap binds (dt = 2 sqrt(3) / max_steps: only on long rays)
    for form in (0, 1):
        with raymarching.count_form(form):
            counter = torch.zeros(2, dtype=torch.int32, device=cuda)
            xyzs, dirs, deltas, rays_t = raymarching.march_rays_train(
                _t(o, cuda), _t(d, cuda), bound, _t(bf, cuda), cas, HG, _t(nears, cuda), _t(fars, cuda), counter, -1, True,
                -1, True, 0, max_steps, _t(noises, cuda))
        m = xyzs.shape[0]
        assert np.array_equal(counter.cpu().numpy(), cr) and np.array_equal(rays_t.cpu().numpy(), rr), (case, form)
        assert np.array_equal(xyzs.cpu().numpy(), xr[:m]) and np.array_equal(deltas.cpu().numpy(), lr[:m]), (case, form)


def test_march_with_unaligned_bitfield(cuda, rays):
    """The 64-bit cached occupancy lookups need an 8-byte aligned bitfield; any other pointer takes the byte-wise
    path.  Same samples either way (and both equal the oracle's, see test_march_train_exact)."""
    from trinerflet_amd import raymarching
    o, d, aabb, nears, fars = rays
    bf = scene.sphere_bitfield(HG, CAS, BOUND, 0.8, 0.7)
    N = o.shape[0]
    noises = _t(np.random.default_rng(5).random(N).astype(np.float32), cuda)
    holder = torch.zeros(bf.size + 8, dtype=torch.uint8, device=cuda)
    outs = []
    for shift in (0, 3):
        view = holder[shift:shift + bf.size]
        view.copy_(_t(bf, cuda))
        assert view.data_ptr() % 8 == (holder.data_ptr() + shift) % 8
        counter = torch.zeros(2, dtype=torch.int32, device=cuda)
        out = raymarching.march_rays_train(_t(o, cuda), _t(d, cuda), BOUND, view, CAS, HG, _t(nears, cuda),
                                           _t(fars, cuda), counter, -1, True, -1, True, 0, 1024, noises)
        outs.append([t.clone() for t in out] + [counter.clone()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def _samples(cuda, rays, seed=0):
    o, d, aabb, nears, fars = rays
    bf = scene.sphere_bitfield(HG, CAS, BOUND, 0.8, 0.0)
    N = o.shape[0]
    noises = np.random.default_rng(5).random(N).astype(np.float32)
    xr, dr, lr, rr, cr = cref.march_rays_train(o, d, BOUND, bf, CAS, HG, nears, fars, noises, N * 1024)
    M = int(cr[0])
    M += 128 - M % 128
    g = np.random.default_rng(seed)
    sig = np.exp(g.standard_normal(M) * 2.0).astype(np.float32)  # wide range: early stops and faint rays
    rgb = g.random((M, 3)).astype(np.float32)
    return sig, rgb, lr[:M], rr, N, M


def test_composite_train_fwd_bwd(cuda, rays):
    from trinerflet_amd import raymarching
    sig, rgb, deltas, rr, N, M = _samples(cuda, rays)
    rr = rr.copy()
    rr[5, 1] = M  # an overflowing ray: outputs must be zero, no gradient
    ws, dep, img = cref.composite_rays_train_forward(sig, rgb, deltas, rr, 1e-4)
    s_t = _t(sig, cuda).requires_grad_(True)
    c_t = _t(rgb, cuda).requires_grad_(True)
    ws_t, dep_t, img_t = raymarching.composite_rays_train(s_t, c_t, _t(deltas, cuda), _t(rr, cuda), 1e-4)
    # fp32 tolerance: wavefront scan vs serial recurrence, __expf on both
    np.testing.assert_allclose(ws_t.detach().cpu().numpy(), ws, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(img_t.detach().cpu().numpy(), img, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(dep_t.detach().cpu().numpy(), dep, rtol=5e-5, atol=1e-5)
    g = np.random.default_rng(1)
    gws = g.standard_normal(N).astype(np.float32)
    gimg = g.standard_normal((N, 3)).astype(np.float32)
    gs, gc = cref.composite_rays_train_backward(gws, gimg, sig, rgb, deltas, rr, ws, img, 1e-4)
    (ws_t * _t(gws, cuda)).sum().add((img_t * _t(gimg, cuda)).sum()).backward()
    np.testing.assert_allclose(c_t.grad.cpu().numpy(), gc, rtol=2e-5, atol=2e-6)
    # grad_sigma has cancellation (T*rgb - (C_final - C)); compare with an absolute scale
    scale = np.abs(gs).max()
    np.testing.assert_allclose(s_t.grad.cpu().numpy(), gs, rtol=1e-3, atol=2e-5 * scale)


def test_composite_backward_covers_every_row_before_the_budget(cuda, rays):
    """With a sample budget that drops the last rays (raymarching.cu:422), every row of the gradient buffers below
    min(M, counter) is written by the backward kernel itself -- kept rays write theirs, the first dropped ray zeroes
    the tail it still owns -- so a caller that ignores the rows behind the count needs no zero fill."""
    from trinerflet_amd import _lib as L
    from trinerflet_amd import raymarching
    o, d, aabb, nears, fars = rays
    bf = scene.sphere_bitfield(HG, CAS, BOUND, 0.8, 0.6)
    N = o.shape[0]
    noises = np.zeros(N, np.float32)
    _, _, _, _, cr = cref.march_rays_train(o, d, BOUND, bf, CAS, HG, nears, fars, noises, N * 1024)
    M = int(cr[0]) * 2 // 3                      # a budget that cuts a ray in the middle of the buffer
    xr, dr, lr, rr, cr = cref.march_rays_train(o, d, BOUND, bf, CAS, HG, nears, fars, noises, M)
    kept = rr[(rr[:, 2] > 0) & (rr[:, 1] + rr[:, 2] <= M)]
    end_kept = int((kept[:, 1] + kept[:, 2]).max())
    assert end_kept < M < int(cr[0])             # the first dropped ray owns [end_kept, M)
    rng = np.random.default_rng(3)
    sig = _t(rng.random(M).astype(np.float32) * 3, cuda)
    rgb = _t(rng.random((M, 3)).astype(np.float32), cuda)
    ws, dep, img = raymarching.composite_rays_train(sig, rgb, _t(lr[:M], cuda), _t(rr, cuda))
    gs = torch.full((M,), float("nan"), device=cuda)
    gc = torch.full((M, 3), float("nan"), device=cuda)
    g_ws = torch.randn(N, device=cuda)
    g_img = torch.randn(N, 3, device=cuda)
    L.check(L.lib().tnl_composite_rays_train_backward(
        L.ptr(g_ws), L.ptr(g_img), L.ptr(sig), L.ptr(rgb), L.ptr(_t(lr[:M], cuda)), L.ptr(_t(rr, cuda)), L.ptr(ws),
        L.ptr(img), L.u32(M), L.u32(N), L.f32(1e-4), L.ptr(gs), L.ptr(gc), L.stream()), "composite_bwd")
    assert torch.isfinite(gs).all() and torch.isfinite(gc).all()
    assert float(gs[end_kept:].abs().sum()) == 0 and float(gc[end_kept:].abs().sum()) == 0
    assert float(gs[:end_kept].abs().sum()) > 0


def test_inference_loop(cuda, rays):
    """run_cuda's eval branch (renderer.py:324-374) with the GPU kernels vs the oracle, same sigma/rgb field."""
    from trinerflet_amd import raymarching
    o, d, aabb, nears, fars = rays
    bf = scene.sphere_bitfield(HG, CAS, BOUND, 0.8, 0.6)
    N = 1024
    o, d, nears, fars = o[:N], d[:N], nears[:N], fars[:N]

    def field(x):  # deterministic analytic field evaluated identically on both sides (float32 numpy)
        s = (8.0 * np.exp(-4.0 * (x ** 2).sum(-1))).astype(np.float32)
        c = (0.5 + 0.5 * np.sin(3.0 * x)).astype(np.float32)
        return s, c

    # oracle
    ws, dep, img = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    alive = np.arange(N, dtype=np.int32)
    rt = nears.copy()
    # gpu
    ws_g, dep_g, img_g = (torch.zeros(N, device=cuda), torch.zeros(N, device=cuda), torch.zeros(N, 3, device=cuda))
    alive_g = torch.arange(N, dtype=torch.int32, device=cuda)
    rt_g = _t(nears, cuda).clone()
    o_g, d_g, bf_g, n_g, f_g = _t(o, cuda), _t(d, cuda), _t(bf, cuda), _t(nears, cuda), _t(fars, cuda)
    step = 0
    while step < 1024:
        n_alive = alive.shape[0]
        assert alive_g.shape[0] == n_alive
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        x, dd, dl = cref.march_rays(n_alive, n_step, alive, rt, o, d, BOUND, bf, CAS, HG, nears, fars,
                                    np.zeros(n_alive, np.float32), 128)
        xg, dg, lg = raymarching.march_rays(n_alive, n_step, alive_g, rt_g, o_g, d_g, BOUND, bf_g, CAS, HG, n_g, f_g,
                                            128, False, 0, 1024)
        assert np.array_equal(xg.cpu().numpy(), x) and np.array_equal(lg.cpu().numpy(), dl)
        s, c = field(x)
        cref.composite_rays(n_alive, n_step, alive, rt, s, c, dl, ws, dep, img, 1e-4)
        raymarching.composite_rays(n_alive, n_step, alive_g, rt_g, _t(s, cuda), _t(c, cuda), lg, ws_g, dep_g, img_g, 1e-4)
        compacted, n_out = raymarching.compact_rays(alive_g)
        alive = alive[alive >= 0]
        assert int(n_out.item()) == alive.shape[0]
        alive_g = compacted[: alive.shape[0]]
        assert np.array_equal(alive_g.cpu().numpy(), alive)        # bit-exact surviving ray ids, in order
        step += n_step
    np.testing.assert_allclose(img_g.cpu().numpy(), img, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ws_g.cpu().numpy(), ws, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dep_g.cpu().numpy(), dep, rtol=1e-4, atol=1e-5)


def test_sh(cuda):
    from trinerflet_amd.shencoder import SHEncoder
    g = np.random.default_rng(2)
    d = g.standard_normal((1000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    enc = SHEncoder(3, 4)
    out = enc(_t(d, cuda))
    np.testing.assert_allclose(out.cpu().numpy(), cref.sh4(d), rtol=1e-6, atol=1e-7)
    # dy_dx path (unused on the hot path): finite differences in float64 of the oracle polynomial
    dt = _t(d, cuda).requires_grad_(True)
    w = _t(g.standard_normal((1000, 16)).astype(np.float32), cuda)
    (enc(dt) * w).sum().backward()
    eps = 1e-3
    num = np.zeros_like(d)
    for k in range(3):
        dp, dm = d.copy(), d.copy()
        dp[:, k] += eps
        dm[:, k] -= eps
        num[:, k] = ((cref.sh4(dp).astype(np.float64) - cref.sh4(dm)) * w.cpu().numpy()).sum(-1) / (2 * eps)
    np.testing.assert_allclose(dt.grad.cpu().numpy(), num, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("degree", [1, 2, 3, 4, 5, 6, 7, 8])
def test_sh_all_degrees_vs_reference_formulas(cuda, golden_dir, degree):
    """tests/golden/sh_reference.npz = the reference kernel's own statements (shencoder.cu:48-122 values, :125-354
    derivatives) evaluated in fp32 by make_golden_sh.py: every degree the reference supports, values and dy_dx."""
    import os
    from trinerflet_amd.shencoder import SHEncoder
    g = np.load(os.path.join(golden_dir, "sh_reference.npz"))
    n = degree * degree
    d = _t(g["dirs"], cuda).requires_grad_(True)
    out = SHEncoder(3, degree)(d)
    want = g["out"][:, :n]
    tol = 1e-6 if degree <= 4 else 2e-5            # <= 4: the same polynomials; above: generated by recurrence
    assert np.abs(out.detach().cpu().numpy() - want).max() <= tol * max(1.0, np.abs(want).max())
    # derivatives through the dy_dx path: VJP with one-hot cotangents per input axis equals the tables' column sums
    w = torch.from_numpy(np.random.default_rng(degree).standard_normal((96, n)).astype(np.float32)).to(cuda)
    (gin,) = torch.autograd.grad(out, d, w)
    ref = np.stack([(g[k][:, :n] * w.cpu().numpy()).sum(-1) for k in ("dx", "dy", "dz")], -1)
    assert np.abs(gin.cpu().numpy() - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.gpu
def test_march_train_record_path_equals_two_march_path(cuda):
    """tnl_march_rays_train writes the samples either by a second march of every ray (minimal workspace) or from the t
    values its count pass recorded (workspace_rec): same bits, including a sample budget that drops rays."""
    import trinerflet_amd._lib as L
    from trinerflet_amd import raymarching, synthetic
    lib = L.lib()
    N, max_steps, Cc, Hg, bound = 3000, 256, 2, 64, 1.5
    rng = np.random.default_rng(3)
    poses = synthetic.hemisphere_poses(4, seed=1)
    pix = np.stack([rng.integers(0, 4, N), rng.integers(0, 800 * 800, N)], -1)
    o, d = (torch.from_numpy(a).to(cuda) for a in synthetic.get_rays(poses, pix))
    bits = torch.from_numpy(synthetic.sphere_bitfield(Hg, Cc, bound, 0.8, 0.0)).to(cuda)
    aabb = torch.tensor([-bound] * 3 + [bound] * 3, device=cuda)
    nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.2)
    noise = torch.from_numpy(rng.random(N).astype(np.float32)).to(cuda)
    outs = []
    for dt_gamma in (0.0, 1.0 / 128):
        for M in (N * 40, 20000):                       # roomy budget / one that drops rays
            res = []
            for rec in (False, True):
                nws = lib.tnl_march_rays_train_workspace_rec(L.u32(N), L.u32(max_steps)) if rec else \
                    lib.tnl_march_rays_train_workspace(L.u32(N))
                ws = torch.empty(nws, dtype=torch.int32, device=cuda)
                # sentinel-filled: the kernels must not rely on the caller's zero fill for rows a dropped ray leaves
                xyzs, dirs = torch.full((M, 3), 7.0, device=cuda), torch.full((M, 3), 7.0, device=cuda)
                deltas = torch.full((M, 2), 7.0, device=cuda)
                rays = torch.empty(N, 3, dtype=torch.int32, device=cuda)
                counter = torch.zeros(2, dtype=torch.int32, device=cuda)
                L.check(lib.tnl_march_rays_train(L.ptr(o), L.ptr(d), L.ptr(bits), L.f32(bound), L.f32(dt_gamma),
                                                 L.u32(max_steps), L.u32(N), L.u32(Cc), L.u32(Hg), L.u32(M), L.ptr(nears),
                                                 L.ptr(fars), L.ptr(xyzs), L.ptr(dirs), L.ptr(deltas), L.ptr(rays),
                                                 L.ptr(counter), L.ptr(noise), L.ptr(ws), L.u32(nws), L.stream()), "march")
                res.append((xyzs, dirs, deltas, rays, counter))
            for a, b in zip(*res):
                assert torch.equal(a, b)
            total = int(res[0][4][0])
            assert total > 0
            if total > M:   # rays were dropped: the kept rays' rows are followed by a zeroed tail, nothing is left over
                rays = res[0][3].cpu().numpy()
                kept = rays[rays[:, 1] + rays[:, 2] <= M]
                end = int((kept[:, 1] + kept[:, 2]).max())
                assert 0 < end < M and not res[0][0][end:].any() and not res[0][2][end:].any()
                assert not (res[0][0][:end] == 7.0).all(1).any()
            else:
                assert bool((res[0][0][total:] == 7.0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("caps", [(0, 0), (3, 2), (1, 1)], ids=["full-width", "capped-3-2", "one-workgroup-each"])
def test_march_fused_tile_count_equals_sort(cuda, caps):
    """march_rays_train(sort=(R, ws)) + plane_grad_sort_counted against plane_grad_sort on the samples it wrote: same
    bin offsets, and per bin the same set of sample ids (the order inside a bin is the atomics' in both); with a budget
    that drops rays as well (the zeroed tail rows are samples of the sort too).  caps: the emit and fill passes launched
    with that many workgroups (raymarching.side_caps: grid-stride over the rays / samples, TrainStep's prefetch) -- the
    same samples to the bit as the full-width launch, the same lists."""
    from trinerflet_amd import _lib as L
    from trinerflet_amd import raymarching, synthetic
    from trinerflet_amd.nerf import field as gfield
    N, max_steps, Cc, Hg, bound, R = 3000, 256, 2, 64, 1.5, 256
    rng = np.random.default_rng(5)
    poses = synthetic.hemisphere_poses(4, seed=2)
    pix = np.stack([rng.integers(0, 4, N), rng.integers(0, 800 * 800, N)], -1)
    o, d = (torch.from_numpy(a).to(cuda) for a in synthetic.get_rays(poses, pix))
    bits = torch.from_numpy(synthetic.sphere_bitfield(Hg, Cc, bound, 0.8, 0.0)).to(cuda)
    aabb = torch.tensor([-bound] * 3 + [bound] * 3, device=cuda)
    nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.2)
    noise = torch.from_numpy(rng.random(N).astype(np.float32)).to(cuda)
    SUBS = 4                                # BIN_SUBS of csrc/bin_common.h: sub-bins per (plane, tile)
    nb = 3 * (R // 32) * (R // 8) * SUBS
    ent0 = 3 * (nb + 1) + 1 + (nb + 1023) // 1024 + 8

    def parse(ws):
        w = ws.view(torch.int32).cpu().numpy()
        offsets = w[nb + 1:2 * nb + 2]
        return offsets, w[ent0:ent0 + offsets[-1]]
    for budget in (40 * N, 20000 - 20000 % 128):
        counter = torch.zeros(2, dtype=torch.int32, device=cuda)
        mc = budget + (128 - budget % 128)
        ws = gfield.plane_grad_sort_workspace(mc, R, cuda)
        with raymarching.side_caps(*caps):
            xyzs, dirs, deltas, rays = raymarching.march_rays_train(o, d, bound, bits, Cc, Hg, nears, fars, counter, budget,
                                                                    True, 128, False, 0, max_steps, noise, False, (R, ws))
            assert xyzs.shape[0] == mc
            gfield.plane_grad_sort_counted(ws, xyzs, bound, R, counter)
        assert L.lib().tnl_march_emit_cap(-1) == 0 and L.lib().tnl_plane_grad_fill_cap(-1) == 0      # restored
        if caps != (0, 0):
            c2 = torch.zeros(2, dtype=torch.int32, device=cuda)
            full = raymarching.march_rays_train(o, d, bound, bits, Cc, Hg, nears, fars, c2, budget, True, 128, False, 0,
                                                max_steps, noise, False, None)
            assert torch.equal(c2, counter)
            m_used = min(int(counter[0]), mc)          # (rows behind the samples are whatever the allocation held)
            for a_, b_ in zip((xyzs, dirs, deltas), full[:3]):
                assert torch.equal(a_[:m_used], b_[:m_used])
            assert torch.equal(rays, full[3])
        ref = gfield.plane_grad_sort(xyzs, bound, R, counter)
        torch.cuda.synchronize()
        (off_a, ent_a), (off_b, ent_b) = parse(ws), parse(ref)
        assert np.array_equal(off_a, off_b) and off_a[-1] >= min(int(counter[0]), mc) * 3
        for b in np.flatnonzero(np.diff(off_a))[::7]:
            assert np.array_equal(np.sort(ent_a[off_a[b]:off_a[b + 1]]), np.sort(ent_b[off_b[b]:off_b[b + 1]]))
        assert np.array_equal(np.sort(ent_a), np.sort(ent_b))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["ball", "shell", "offset_box", "touching_boundary", "empty"])
def test_far_clipped_to_the_occupied_box_gives_the_same_samples(cuda, rays, shape):
    """tnl_occupied_box + tnl_clip_fars: the training march with far = min(far, exit of the occupied cells' box) against
    the march to the real far -- identical rays / samples / counts -- for several occupancies (a box away from the
    centre, cells on the volume boundary where positions are clamped, nothing occupied at all)."""
    from trinerflet_amd import raymarching
    o, d, aabb, nears, fars = rays
    if shape in ("ball", "shell"):
        bf = scene.sphere_bitfield(HG, CAS, BOUND, 0.8, 0.7 if shape == "shell" else 0.0)
    else:
        grid = np.zeros((CAS, HG, HG, HG), bool)
        if shape == "offset_box":
            grid[0, 70:100, 20:50, 60:90] = True
            grid[1, 40:50, 80:90, 64:70] = True
        elif shape == "touching_boundary":
            grid[0, 100:128, 50:80, 0:30] = True            # cascade 0 spans [-1, 1]: its last cells are interior of the volume
            grid[1, 118:128, 60:70, 60:70] = True           # cascade 1 reaches the volume boundary (x = +1.5)
        bf = np.zeros((CAS, HG ** 3 // 8), np.uint8)
        ax = np.arange(HG)
        cells = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
        mort = cref.morton3D(cells.astype(np.int32))
        for c in range(CAS):
            flat = np.zeros(HG ** 3, bool)
            flat[mort] = grid[c].reshape(-1)
            bf[c] = np.packbits(flat, bitorder="little")
        bf = bf.reshape(-1)
    N = o.shape[0]
    noises = _t(np.random.default_rng(5).random(N).astype(np.float32), cuda)
    bits = _t(bf, cuda)
    box = raymarching.occupied_box(bits, CAS, HG, BOUND)
    fars_c = raymarching.clip_fars(_t(o, cuda), _t(d, cuda), _t(fars, cuda), box)
    assert bool((fars_c <= _t(fars, cuda)).all())
    outs = []
    for f in (_t(fars, cuda), fars_c):
        counter = torch.zeros(2, dtype=torch.int32, device=cuda)
        out = raymarching.march_rays_train(_t(o, cuda), _t(d, cuda), BOUND, bits, CAS, HG, _t(nears, cuda), f, counter,
                                           -1, True, -1, True, 0, 1024, noises)
        outs.append([t.clone() for t in out] + [counter.clone()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    total = int(outs[0][4][0])
    assert (total == 0) == (shape == "empty")
    if shape != "empty":
        assert float((fars_c < _t(fars, cuda)).float().mean()) > 0.5      # the clipping is real
