"""TrainStep(graph=True): the steady-state steps of a density-grid period replayed as captured HIP graphs must be the steps
TrainStep launches eagerly -- same losses, same parameters, same occupancy grid, bit for bit (ordered plane-gradient
reduction so that the eager run is reproducible) -- over several periods, with real refreshes in between, and a change of
the occupancy window (the captured steps are dropped and captured again)."""
import copy

import numpy as np
import pytest
import torch

from trinerflet_amd import synthetic

pytestmark = pytest.mark.gpu


def _model(cuda, R=512, C=16):
    from trinerflet_amd.nerf.network import NeRFNetwork
    m = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10,
                    bg_radius=-1, hidden_dim=64, hidden_dim_color=64, triplane_channels=C, triplane_resolution=R,
                    triplane_wavelet_levels=8, wavelet_type="bior6.8").to(cuda)
    synthetic.init_field_parameters(m, seed=0)
    return m


def _batches(cuda, n, N):
    out = []
    for b in range(n):
        o, d = synthetic.training_rays(N, n_cams=16, seed=10 + b, H=200, W=200)
        rng = np.random.default_rng(b)
        out.append(tuple(torch.from_numpy(a).to(cuda) for a in (o, d, synthetic.target_colors(d), rng.random(N).astype(np.float32))))
    return out


def _run(cuda, graph, steps, radii):
    from trinerflet_amd.train import TrainStep
    m = _model(cuda)
    torch.manual_seed(5)                      # the refreshes' jitter (torch.rand_like in update_extra_state)
    ts = TrainStep(m, lr=1e-2, wavelet_regularization=0.2, iters=1000, fp16=True, background_color=0.0, deterministic=True,
                   defer_adam=True, graph=graph)
    state = {"k": 0}

    def reimpose():      # an analytic occupancy after every refresh; its radius (hence the window) changes along `radii`
        r = radii[min(state["k"], len(radii) - 1)]
        state["k"] += 1
        m.density_bitfield.copy_(torch.from_numpy(synthetic.sphere_bitfield(128, m.cascade, 1.5, r, 0.0)).to(cuda))
        m.mean_count = 30000
    ts.post_refresh = reimpose
    batches = _batches(cuda, 4, 4096)
    losses = []
    for k in range(steps):
        o, d, gt, nz = batches[k % 4]
        o2, d2, _, nz2 = batches[(k + 1) % 4]
        losses.append(ts.step(o, d, gt, noises=nz, next_rays=(o2, d2, nz2)).clone())
    ts.flush_deferred()
    torch.cuda.synchronize()
    params = {k_: v.detach().clone() for k_, v in m.named_parameters()}
    return torch.stack(losses).cpu().numpy(), params, m.density_grid.clone(), ts


def test_captured_steps_are_the_eager_steps(cuda):
    radii = [0.5, 0.5, 0.5, 0.7, 0.7]            # periods 0-2 one window, then a larger one
    steps = 16 * 5 + 3
    l0, p0, g0, ts0 = _run(cuda, False, steps, radii)
    l1, p1, g1, ts1 = _run(cuda, True, steps, radii)
    assert ts0.graph_replays == 0
    # positions 1..14 of every period after the first refresh are replayed; each window captures its 14 graphs once
    assert ts1.graph_captures == 28 and ts1.graph_replays >= 14 * 4, (ts1.graph_captures, ts1.graph_replays)
    # (the reported loss sums the rays' squared errors with float atomics: equal to rounding, not to the bit)
    assert np.allclose(l0, l1, rtol=2e-6, atol=0), np.abs(l0 - l1).max()
    assert torch.equal(g0, g1)
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k


def test_a_step_outside_the_pattern_runs_eagerly(cuda):
    """No announced next batch, a batch other than the announced one: such steps fall back to the eager launches and the
    captured ones resume afterwards, still bit for bit."""
    from trinerflet_amd.train import TrainStep
    outs = []
    for graph in (False, True):
        m = _model(cuda)
        torch.manual_seed(5)
        ts = TrainStep(m, lr=1e-2, wavelet_regularization=0.2, iters=1000, fp16=True, deterministic=True, defer_adam=True,
                       graph=graph)
        bf = torch.from_numpy(synthetic.sphere_bitfield(128, m.cascade, 1.5, 0.5, 0.0)).to(cuda)

        def reimpose(m=m, bf=bf):
            m.density_bitfield.copy_(bf)
            m.mean_count = 30000
        ts.post_refresh = reimpose
        b = _batches(cuda, 4, 4096)
        losses = []
        for k in range(40):
            o, d, gt, nz = b[k % 4]
            nxt = b[(k + 1) % 4]
            if k == 21:
                nxt_arg = None                                   # nothing announced
            elif k == 25:
                nxt_arg = (b[(k + 2) % 4][0], b[(k + 2) % 4][1], b[(k + 2) % 4][3])   # another batch than the one that comes
            else:
                nxt_arg = (nxt[0], nxt[1], nxt[3])
            losses.append(ts.step(o, d, gt, noises=nz, next_rays=nxt_arg).clone())
        ts.flush_deferred()
        outs.append((torch.stack(losses).cpu().numpy(), {k_: v.detach().clone() for k_, v in m.named_parameters()}, ts))
    (l0, p0, _), (l1, p1, ts1) = outs
    assert ts1.graph_replays > 10
    assert np.allclose(l0, l1, rtol=2e-6, atol=0)
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k


def test_graphs_that_are_recaptured_every_period_switch_themselves_off(cuda):
    """On a real trajectory every density-grid refresh moves the sample budget (model.mean_count is part of the captured
    launches' signature), so every steady-state step would be captured and replayed once -- never faster than the eager
    step.  After three signatures in a row without a second replay TrainStep(graph=True) warns, drops
    its graphs and continues eagerly; the training is the eager one bit for bit either way."""
    from trinerflet_amd.train import TrainStep
    outs = []
    for graph in (False, True):
        m = _model(cuda)
        torch.manual_seed(5)
        ts = TrainStep(m, lr=1e-2, wavelet_regularization=0.2, iters=1000, fp16=True, deterministic=True, defer_adam=True,
                       graph=graph)
        bf = torch.from_numpy(synthetic.sphere_bitfield(128, m.cascade, 1.5, 0.5, 0.0)).to(cuda)
        state = {"k": 0}

        def reimpose(m=m, bf=bf, state=state):
            m.density_bitfield.copy_(bf)
            state["k"] += 1
            m.mean_count = 30000 + 128 * state["k"]            # a budget that moves with every refresh
        ts.post_refresh = reimpose
        b = _batches(cuda, 4, 4096)
        losses = []
        import warnings
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            for k in range(16 * 6 + 2):
                o, d, gt, nz = b[k % 4]
                nxt = b[(k + 1) % 4]
                losses.append(ts.step(o, d, gt, noises=nz, next_rays=(nxt[0], nxt[1], nxt[3])).clone())
        ts.flush_deferred()
        outs.append((torch.stack(losses).cpu().numpy(), {k_: v.detach().clone() for k_, v in m.named_parameters()}, ts,
                     [str(w.message) for w in caught]))
    (l0, p0, _, _), (l1, p1, ts1, msgs) = outs
    assert ts1.graph_auto_disabled and not ts1.graph and any("graphs switched off" in m_ for m_ in msgs), msgs
    assert 14 * 3 <= ts1.graph_captures <= 14 * 4 and ts1.graph_replays == ts1.graph_captures
    assert np.allclose(l0, l1, rtol=2e-6, atol=0)
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k
