"""TrainStep's mode flags in combination ("eight mode flags whose cross-product is only sparsely tested").

Every one of these switches changes HOW the step is executed, none WHAT it computes: with the ordered plane-gradient
reduction (deterministic=True) all of them are documented as bit-identical to the plain step -- the occupancy window
(use_roi), the live / deferred optimiser split (defer_adam), its band pieces (live_bands), the next batch's march on the
side stream (overlap_march) with its two start positions (prefetch_at) and count-pass forms (side_count_form), the far
clip of the in-order march (clip_far_in_order), the optimiser inside the adjoint's column-walk levels (fuse_live), the launch width of the prefetched march's wide passes (side_caps), the banded plane-gradient exchange (overlap_exchange) and the captured
steps (graph).  Here a seeded sample of their cross-product (plus every single flag flipped on its own) trains the same
small model over three density-grid periods -- refresh steps, window changes of the re-imposed occupancy and ring flushes
inside -- and must end on the SAME BITS as the default configuration, parameters and Adam moments alike."""
import random

import numpy as np
import pytest
import torch

from trinerflet_amd import synthetic

pytestmark = pytest.mark.gpu

FLAGS = {            # name -> (default, alternative)
    "use_roi": (True, False), "defer_adam": (True, False), "live_bands": (True, False), "overlap_march": (True, False),
    "prefetch_at": ("bwd", "adjoint"), "side_count_form": (1, 0), "clip_far_in_order": (True, False),
    "overlap_exchange": (0, 3), "graph": (False, True), "fuse_live": (True, False), "side_caps": ("device", (0, 0)),
}


def _combos():
    names = list(FLAGS)
    out = [{}] + [{n: FLAGS[n][1]} for n in names]                  # the default, then every flag on its own
    rng = random.Random(5)
    seen = {tuple(sorted(c.items())) for c in out}
    while len(out) < 1 + len(names) + 10:                            # + ten random mixtures
        c = {n: FLAGS[n][1] for n in names if rng.random() < 0.5}
        if c.get("graph") and c.get("overlap_exchange"):
            c.pop("overlap_exchange")                                # (a captured step takes the single reduction)
        key = tuple(sorted(c.items()))
        if key not in seen:
            seen.add(key)
            out.append(c)
    return out


COMBOS = _combos()


def _run(cuda, combo):
    from trinerflet_amd.nerf.network import NeRFNetwork
    from trinerflet_amd.train import TrainStep
    torch.manual_seed(0)
    m = NeRFNetwork(encoding="triplane_wavelet", bound=1.0, cuda_ray=True, density_thresh=10, hidden_dim=64,
                    hidden_dim_color=64, triplane_channels=16, triplane_resolution=512, triplane_wavelet_levels=8,
                    wavelet_type="bior6.8").to(cuda)
    synthetic.init_field_parameters(m, seed=3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    bfs = [t(synthetic.sphere_bitfield(128, 1, 1.0, r, 0.0)) for r in (0.3, 0.2, 0.25)]      # the window moves at refreshes
    m.density_bitfield.copy_(bfs[0])
    ctor = {k: v for k, v in combo.items() if k in ("use_roi", "defer_adam", "live_bands", "overlap_exchange", "graph")}
    ctor.setdefault("defer_adam", True)      # (the constructor's own default turns the split on from 32 M coefficients)
    ts = TrainStep(m, lr=1e-2, wavelet_regularization=0.2, iters=200, update_extra_interval=8, deterministic=True, **ctor)
    for k in ("overlap_march", "prefetch_at", "side_count_form", "clip_far_in_order", "fuse_live", "side_caps"):
        if k in combo:
            setattr(ts, k, combo[k])
    if "side_caps" not in combo:
        ts.side_caps = (3, 2)        # (the device's default caps exceed this small batch's grids: a few workgroups walk it here)
    period = {"n": 0}

    def reimpose():
        period["n"] += 1
        m.density_bitfield.copy_(bfs[period["n"] % 3])
        m.mean_count = 600000
    ts.post_refresh = reimpose
    m.mean_count = 600000
    batches = []
    for b in range(4):
        o, d = synthetic.training_rays(4096, n_cams=6, seed=20 + b)
        nz = np.random.default_rng(b).random(4096).astype(np.float32)
        batches.append((t(o), t(d), t(synthetic.target_colors(d)), t(nz)))
    losses = []
    fused_seen = set()
    for k in range(26):                                   # refreshes at 0, 8, 16, 24
        o, d, gt, nz = batches[k % 4]
        nxt = batches[(k + 1) % 4]
        ts.step(o, d, gt, noises=nz, next_rays=(nxt[0], nxt[1], nxt[3]))
        fused_seen.update(ts._fused_levels)
        losses.append(float(ts.last["mse"]))      # (the step's loss also carries the L1 value, whose deferred share is reported apart)
    ts.flush_deferred()
    ts.sync_sharded_parameters()
    return (losses, [p.detach().clone() for p in m.parameters()], ts.coef.m.clone(), ts.coef.v.clone(), ts.ll.m.clone(),
            {"deferred": ts.deferred_steps, "replays": getattr(ts, "graph_replays", 0), "fused": fused_seen})


@pytest.fixture(scope="module", autouse=True)
def walk_levels(cuda):
    """The 128-, 64- and 32-coefficient levels of this small model on the column-walk kernels (production: n >= 512), so that
    the adjoint levels that carry the optimiser (fuse_live) are part of every configuration, the baseline's included."""
    from trinerflet_amd import _lib as L
    L.lib().tnl_idwt_set_walk_min_n(L.u32(32))
    yield
    L.lib().tnl_idwt_set_walk_min_n(L.u32(0))


@pytest.fixture(scope="module")
def baseline(cuda, walk_levels):
    out = _run(cuda, {})
    assert out[5]["deferred"] > 8          # the geometry exercises the live / deferred split
    assert len(out[5]["fused"]) >= 2       # ... and adjoint levels with the optimiser in their epilogue
    return out


@pytest.mark.parametrize("idx", range(1, len(COMBOS)))
def test_flag_combination_ends_on_the_default_configuration_s_bits(cuda, baseline, idx):
    combo = COMBOS[idx]
    got = _run(cuda, combo)
    np.testing.assert_allclose(got[0], baseline[0], rtol=3e-6, err_msg=str(combo))      # (reported losses: float-atomic sums)
    for a, b in zip(got[1], baseline[1]):
        assert torch.equal(a, b), (combo, a.shape)
    for k in (2, 3, 4):
        assert torch.equal(got[k], baseline[k]), (combo, k)
    if (combo.get("graph") and combo.get("overlap_march", True) and combo.get("use_roi", True) and combo.get("defer_adam", True)
            and not combo.get("overlap_exchange")):
        assert got[5]["replays"] > 0, combo        # (a capture needs the window, the deferred split and the prefetch)
    if combo.get("defer_adam", True) and combo.get("use_roi", True):
        assert got[5]["deferred"] > 0, combo


def test_the_matrix_covers_every_flag_both_ways():
    for n, (dflt, alt) in FLAGS.items():
        assert any(c.get(n, dflt) == alt for c in COMBOS) and any(c.get(n, dflt) == dflt for c in COMBOS), n
    assert len(COMBOS) == len({tuple(sorted(c.items())) for c in COMBOS}) >= 20
    assert sum(len(c) >= 3 for c in COMBOS) >= 6              # real mixtures, not only single flips


def test_a_skipped_step_in_the_steady_state_moves_nothing(cuda, walk_levels):
    """GradScaler's verdict reaches the adjoint levels that carry the optimiser (fuse_live) through the step record: a step
    whose gradients overflow leaves every coefficient, moment and MLP weight where it was -- the live pieces the fused levels
    would have updated included -- and halves the scale; the step after it trains on."""
    from trinerflet_amd.nerf.network import NeRFNetwork
    from trinerflet_amd.train import TrainStep
    torch.manual_seed(0)
    m = NeRFNetwork(encoding="triplane_wavelet", bound=1.0, cuda_ray=True, density_thresh=10, hidden_dim=64,
                    hidden_dim_color=64, triplane_channels=16, triplane_resolution=512, triplane_wavelet_levels=8,
                    wavelet_type="bior6.8").to(cuda)
    synthetic.init_field_parameters(m, seed=3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    m.density_bitfield.copy_(t(synthetic.sphere_bitfield(128, 1, 1.0, 0.3, 0.0)))
    ts = TrainStep(m, lr=1e-2, wavelet_regularization=0.2, iters=200, update_extra_interval=0, defer_adam=True)
    m.mean_count = 600000
    o, d = synthetic.training_rays(4096, n_cams=6, seed=20)
    o, d, gt = t(o), t(d), t(synthetic.target_colors(d))
    for _ in range(3):
        ts.step(o, d, gt)
    assert len(ts._fused_levels) >= 1 and float(ts.last["found_inf"]) == 0.0          # the steady state, fused
    snap = lambda: ([p.detach().clone() for p in m.parameters()], ts.coef.m.clone(), ts.coef.v.clone(), ts.ll.m.clone(),
                    ts.mlp.m.clone())
    before = snap()
    scale = float(ts.scale)
    ts.scale.fill_(2.0 ** 60)                             # this step's fp16 gradients overflow
    ts.step(o, d, gt)
    assert float(ts.last["found_inf"]) == 1.0 and len(ts._fused_levels) >= 1
    assert float(ts.scale) == 2.0 ** 59
    after = snap()
    for a, b in zip(before[0], after[0]):
        assert torch.equal(a, b)
    for a, b in zip(before[1:], after[1:]):
        assert torch.equal(a, b)
    ts.scale.fill_(scale)
    ts.step(o, d, gt)
    assert float(ts.last["found_inf"]) == 0.0
    assert not torch.equal(after[1], ts.coef.m)            # ... and the next step moves them again
    ts.flush_deferred()
