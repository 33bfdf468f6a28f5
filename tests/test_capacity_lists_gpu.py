"""One-pass tile lists ("capacity lists", include/trinerflet_hip.h): the march writes the plane-gradient tile lists into
fixed spans sized from an earlier batch's counts, entries that do not fit spill and are added with float atomics -- the
plane gradient must be that of the counting sort + tile reduction (tests/test_field_gpu.py pins that one to the oracle;
semantics: grid_sampler_2d_backward's scatter, triplane_encoder.py:329) up to fp32 summation order, WHATEVER the spans
were sized from: the same batch (nothing spills), a much smaller batch (a lot spills), no batch at all (everything that
exceeds 16 entries per sub-bin spills)."""
import copy

import numpy as np
import pytest
import torch

from trinerflet_amd import synthetic

pytestmark = pytest.mark.gpu

BOUND = 1.5


def _rays(cuda, N, seed):
    o, d = synthetic.training_rays(N, n_cams=4, seed=seed)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    return t(o), t(d), t(np.random.default_rng(seed).random(N).astype(np.float32))


def _march(cuda, o, d, nz, bf, M, sort):
    from trinerflet_amd import raymarching
    aabb = torch.tensor([-BOUND] * 3 + [BOUND] * 3, dtype=torch.float32, device=cuda)
    nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.2)
    counter = torch.zeros(2, dtype=torch.int32, device=cuda)
    x, dd, dl, rr = raymarching.march_rays_train(o, d, BOUND, bf, 2, 128, nears, fars, counter, M, True, 128, False, 0, 1024,
                                                 nz, True, sort)
    return x, counter


@pytest.mark.parametrize("C,R", [(16, 256), (32, 512), (48, 256)])
@pytest.mark.parametrize("source", ["same_batch", "small_batch", "empty"])
@pytest.mark.parametrize("windowed", [False, True])
def test_capacity_lists_give_the_counting_sorts_plane_gradient(cuda, C, R, source, windowed):
    from trinerflet_amd import occupancy
    from trinerflet_amd.nerf import field as F_
    N = 3000
    bf_np = synthetic.sphere_bitfield(128, 2, BOUND, 0.5, 0.0)
    bf = torch.from_numpy(bf_np).to(cuda)
    o, d, nz = _rays(cuda, N, 3)
    # the batch's sample count (its budget: the next multiple of 128, the wrapper's rule)
    x0, cnt = _march(cuda, o, d, nz, bf, -1, None)
    total = int(cnt[0])
    assert total > 50 * N // 10
    budget = total + 64
    mc = budget + (128 - budget % 128)
    # reference: the march that counts + the counting sort + the tile reduction
    ws = F_.plane_grad_sort_workspace(mc, R, cuda)
    x, counter = _march(cuda, o, d, nz, bf, budget, (R, ws))
    assert x.shape[0] == mc and int(counter[0]) == total
    F_.plane_grad_sort_counted(ws, x, BOUND, R, counter)
    g = torch.Generator(device=cuda).manual_seed(1)
    dfeat = torch.randn(3, mc, C, generator=g, device=cuda).to(torch.float16)
    roi = occupancy.window(bf, 2, 128, BOUND, R) if windowed else None
    assert (roi is not None) == windowed
    roi10 = None if roi is None else list(roi) + [C, 0]
    shape = (3, C, R, R) if roi is None else (3 * C, roi[7], roi[6])
    want = torch.full(shape, float("nan"), device=cuda)
    F_.plane_grad_reduce(ws, dfeat, x, BOUND, C, R, want, channel_major=True, roi=roi10)
    # the spans' source
    if source == "same_batch":
        src = ws
    elif source == "small_batch":
        o2, d2, nz2 = _rays(cuda, 200, 9)
        src = F_.plane_grad_sort_workspace(mc, R, cuda)
        _march(cuda, o2, d2, nz2, bf, budget, (R, src))             # counts only: that is all the table reads
    else:
        src = torch.zeros_like(ws)
    table = F_.plane_grad_capacity_table(src, R, mc)
    cws = F_.plane_grad_capacity_workspace(mc, R, cuda)
    cws.view(torch.int32)[: cws.numel() // 4].fill_(0x7f7f7f7f)      # stale contents must not matter
    x2, counter2 = _march(cuda, o, d, nz, bf, budget, (R, cws, table))
    assert torch.equal(x2, x) and torch.equal(counter2, counter)    # the same samples
    got = torch.full(shape, float("nan"), device=cuda)
    flag = torch.zeros(1, dtype=torch.int32, device=cuda)
    F_.plane_grad_reduce(cws, dfeat, x2, BOUND, C, R, got, channel_major=True, roi=roi10, capacity=True, nonfinite_flag=flag)
    spilled = int(F_.plane_grad_capacity_spilled(cws, mc, R))
    entries = 3.4 * total
    if source == "same_batch":
        assert spilled == 0, spilled
    else:
        assert spilled > (0.3 if source == "small_batch" else 0.02) * entries, (spilled, entries)
    assert int(flag) == 0 and bool(torch.isfinite(got).all())
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    assert err < 2e-5 * scale, (source, err, scale, spilled)
    # a non-finite feature gradient is reported by the spill path as by the tile kernel
    if source == "empty" and not windowed:
        dfeat2 = dfeat.clone()
        dfeat2[:, :total] = float("inf")
        flag.zero_()
        F_.plane_grad_reduce(cws, dfeat2, x2, BOUND, C, R, got, channel_major=True, roi=roi10, capacity=True, nonfinite_flag=flag)
        assert int(flag) == 1


def test_capacity_workspace_of_another_size_is_refused(cuda):
    from trinerflet_amd.nerf import field as F_
    R, C, mc = 256, 16, 128 * 200
    bf = torch.from_numpy(synthetic.sphere_bitfield(128, 2, BOUND, 0.5, 0.0)).to(cuda)
    o, d, nz = _rays(cuda, 300, 3)
    ws = F_.plane_grad_sort_workspace(mc, R, cuda)
    x, counter = _march(cuda, o, d, nz, bf, mc - 1, (R, ws))
    F_.plane_grad_sort_counted(ws, x, BOUND, R, counter)
    table = F_.plane_grad_capacity_table(ws, R, mc)
    cws = F_.plane_grad_capacity_workspace(mc, R, cuda)
    x2, _ = _march(cuda, o, d, nz, bf, mc - 1, (R, cws, table))
    dfeat = torch.zeros(3, mc, C, dtype=torch.float16, device=cuda)
    out = torch.empty(3, C, R, R, device=cuda)
    with pytest.raises(RuntimeError):      # a capacity workspace read as a counting-sort workspace: other layout
        F_.plane_grad_reduce(cws, dfeat, x2, BOUND, C, R, out, channel_major=True)
    with pytest.raises(RuntimeError):
        F_.plane_grad_reduce(ws, dfeat, x, BOUND, C, R, out, channel_major=True, capacity=True)


def test_training_with_capacity_lists_equals_training_with_the_counting_sort(cuda):
    """TrainStep: the prefetched marches of a density-grid period fill capacity lists (spans from the period's first
    batch), nothing else changes -- the same samples, the same training up to what two runs of ONE configuration differ by
    (the tile lists' order follows atomics either way)."""
    from trinerflet_amd.nerf.network import NeRFNetwork
    from trinerflet_amd.train import TrainStep
    N, bound = 2048, 1.0
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    batches = []
    for seed in (7, 8, 9):
        o, d = synthetic.training_rays(N, n_cams=4, seed=seed)
        batches.append((t(o), t(d), t(synthetic.target_colors(d)), t(np.random.default_rng(seed).random(N).astype(np.float32))))
    base = NeRFNetwork(encoding="triplane_wavelet", bound=bound, cuda_ray=True, density_thresh=10, hidden_dim=64,
                       hidden_dim_color=64, triplane_channels=16, triplane_resolution=256, triplane_wavelet_levels=4,
                       wavelet_type="bior6.8").to(cuda)
    synthetic.init_field_parameters(base, seed=3)
    bf = t(synthetic.sphere_bitfield(128, 1, bound, 0.4, 0.0))
    base.density_bitfield.copy_(bf)
    res = {}
    for tag in ("sort", "capacity", "sort2"):
        m = copy.deepcopy(base)
        ts = TrainStep(m, update_extra_interval=4)
        ts.capacity_lists = tag == "capacity"
        ts.post_refresh = lambda m=m: m.density_bitfield.copy_(bf)
        m.mean_count = 0
        losses, counts = [], []
        for it in range(10):
            o, d, gt, nz = batches[it % 3]
            no, nd, _, nnz = batches[(it + 1) % 3]
            losses.append(float(ts.step(o, d, gt, noises=nz, next_rays=(no, nd, nnz))))
            counts.append(int(ts.last["counter"][0]))
        torch.cuda.synchronize()
        res[tag] = (losses, [p.detach().clone() for p in m.parameters()], counts, ts)
    tsc = res["capacity"][3]
    # (steps 0, 4, 8 refresh: in-order march + counting sort + a new table; step 1 has no budget yet: mean_count = 0)
    assert tsc.capacity_marches >= 4 and tsc.capacity_tables >= 2, (tsc.capacity_marches, tsc.capacity_tables)
    assert res["sort"][3].capacity_marches == 0
    assert res["capacity"][2] == res["sort"][2]
    np.testing.assert_allclose(res["sort"][0], res["capacity"][0], rtol=2e-4)
    for a, b, c in zip(res["sort"][1], res["capacity"][1], res["sort2"][1]):
        far = lambda x, y: int(((x - y).abs() > 2e-3 + 1e-3 * y.abs()).sum())
        assert far(a, b) <= 3 * far(a, c) + max(8, int(1e-4 * a.numel())) and float((a - b).abs().max()) < 6e-2
