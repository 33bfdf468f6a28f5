"""trinerflet_amd.optim.FusedAdamL1 against torch.optim.Adam (reconstruction/main_nerf.py:119: Adam(betas=(0.9, 0.99),
eps=1e-15)) driven the way the reference's loop drives it (utils.py:1166-1173: scaler.scale(loss).backward();
scaler.step(optimizer); scaler.update()), including a skipped step, a weight-decay group, the checkpoint layout in both
directions, and the L1 term folded into the pass."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(3, 8, 3, 64, 64), (3, 8, 16, 16), (64, 24), (1001,), (7,)]
    return [torch.nn.Parameter((torch.randn(s, generator=g) * 0.1).to(dev)) for s in shapes]


def _groups(ps, wd=0.0):
    return [{"params": ps[:2], "lr": 1e-2}, {"params": ps[2:], "lr": 3e-3, **({"weight_decay": wd} if wd else {})}]


def _loss(ps, k):
    return sum(((p * (1.0 + 0.1 * k)) ** 2).sum() * (0.5 + i) + (p.sin() * (k + 1)).sum() for i, p in enumerate(ps))


@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_fused_adam_follows_torch_adam_under_gradscaler(cuda, wd):
    from trinerflet_amd.optim import FusedAdamL1
    pa, pb = _params(cuda), _params(cuda)
    oa = torch.optim.Adam(_groups(pa, wd), betas=(0.9, 0.99), eps=1e-15)
    ob = FusedAdamL1(_groups(pb, wd), betas=(0.9, 0.99), eps=1e-15)
    sa, sb = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=4), torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=4)
    for k in range(12):
        for ps, opt, sc in ((pa, oa, sa), (pb, ob, sb)):
            opt.zero_grad(set_to_none=True)
            loss = _loss(ps, k)
            if k == 5:
                loss = loss + ps[3][0] * float("inf")       # a non-finite gradient: the step must be skipped, the scale halved
            sc.scale(loss).backward()
            sc.step(opt)
            sc.update()
        assert float(sa.get_scale()) == float(sb.get_scale()), k
    for a, b in zip(pa, pb):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-5, atol=2e-7)
    for a, b in zip(pa, pb):
        sta, stb = oa.state[a], ob.state[b]
        assert float(sta["step"]) == float(stb["step"]) == 11.0          # one of the twelve was skipped
        np.testing.assert_allclose(stb["exp_avg"].cpu().numpy(), sta["exp_avg"].cpu().numpy(), rtol=2e-5, atol=1e-9)
        np.testing.assert_allclose(stb["exp_avg_sq"].cpu().numpy(), sta["exp_avg_sq"].cpu().numpy(), rtol=2e-5, atol=1e-12)


def test_state_dict_is_torch_adams_in_both_directions(cuda):
    from trinerflet_amd.optim import FusedAdamL1
    pa, pb = _params(cuda, 1), _params(cuda, 1)
    oa = torch.optim.Adam(_groups(pa), betas=(0.9, 0.99), eps=1e-15)
    ob = FusedAdamL1(_groups(pb), betas=(0.9, 0.99), eps=1e-15)
    for k in range(3):
        for ps, opt in ((pa, oa), (pb, ob)):
            opt.zero_grad()
            _loss(ps, k).backward()
            opt.step()
    sda, sdb = oa.state_dict(), ob.state_dict()
    assert set(sda["state"][0]) == set(sdb["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    assert set(sda["param_groups"][0]) <= set(sdb["param_groups"][0])
    # torch -> fused and fused -> torch, then three more steps each: all four trajectories agree
    oa2 = torch.optim.Adam(_groups(pc := [torch.nn.Parameter(p.detach().clone()) for p in pb]), betas=(0.9, 0.99), eps=1e-15)
    oa2.load_state_dict(copy.deepcopy(sdb))
    ob2 = FusedAdamL1(_groups(pd := [torch.nn.Parameter(p.detach().clone()) for p in pa]), betas=(0.9, 0.99), eps=1e-15)
    ob2.load_state_dict(copy.deepcopy(sda))
    for k in range(3, 6):
        for ps, opt in ((pa, oa), (pb, ob), (pc, oa2), (pd, ob2)):
            opt.zero_grad()
            _loss(ps, k).backward()
            opt.step()
    for a, b, c, d in zip(pa, pb, pc, pd):
        for other in (b, c, d):
            np.testing.assert_allclose(other.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-5, atol=2e-7)
    assert float(ob2.state[pd[0]]["step"]) == 6.0


def test_l1_inside_the_pass_equals_the_regulariser_through_autograd(cuda):
    """utils.py:639-655: loss += lam * mean |coef| -- its gradient lam / n * sign(coef) added inside the Adam pass."""
    from trinerflet_amd.optim import FusedAdamL1
    pa, pb = _params(cuda, 2)[:1], _params(cuda, 2)[:1]
    lam = 0.4
    oa = torch.optim.Adam(pa, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    ob = FusedAdamL1(pb, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, l1=lam / pb[0].numel())
    for k in range(5):
        oa.zero_grad()
        (_loss(pa, k) + lam * pa[0].abs().mean()).backward()
        oa.step()
        ob.zero_grad()
        _loss(pb, k).backward()
        ob.step()
    np.testing.assert_allclose(pb[0].detach().cpu().numpy(), pa[0].detach().cpu().numpy(), rtol=2e-5, atol=2e-7)


def test_rejects_what_it_does_not_implement(cuda):
    from trinerflet_amd.optim import FusedAdamL1
    with pytest.raises(ValueError):
        FusedAdamL1(_params(cuda), amsgrad=True)
    p = [torch.nn.Parameter(torch.zeros(8, dtype=torch.float64, device=cuda))]
    opt = FusedAdamL1(p)
    p[0].grad = torch.ones_like(p[0])
    with pytest.raises(ValueError):
        opt.step()


def test_wavelet_feature_views_fuse_abs_mean_and_behave_like_the_parameters(cuda):
    """get_wavelet_features() hands out views of the coefficient parameters whose `.abs().mean()` (utils.py:639-655) is
    one fused pass: same value, same gradient in the PARAMETER's .grad; every other use behaves like the plain tensor."""
    from trinerflet_amd.nerf.network import NeRFNetwork
    m = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, hidden_dim=64, hidden_dim_color=64,
                    triplane_channels=16, triplane_resolution=256, triplane_wavelet_levels=4).to(cuda)
    enc = m.encoder
    g = torch.Generator(device=cuda).manual_seed(0)
    with torch.no_grad():
        for p in enc.planes_features_wavelet_coefs:
            p.copy_(torch.randn(p.shape, generator=g, device=cuda) * 0.05)
            p.view(-1)[::7] = 0.0                         # exact zeros: sign(0) = 0
    views = enc.get_wavelet_features()
    assert len(views) == len(enc.planes_features_wavelet_coefs) == 2 and all(v.numel() == p.numel() and v.shape == p.shape
                                                                           for v, p in zip(views, enc.planes_features_wavelet_coefs))
    tot = sum(v.numel() for v in views)
    reg = sum(v.abs().mean() * (v.numel() / tot) for v in views) / len(views)            # the reference's expression
    (reg * 1024.0).backward()
    got = [p.grad.clone() for p in enc.planes_features_wavelet_coefs]
    enc.zero_grad()
    enc.fused_l1_views = False
    plain = enc.get_wavelet_features()
    assert all(type(v) is torch.nn.Parameter for v in plain)
    reg2 = sum(v.abs().mean() * (v.numel() / tot) for v in plain) / len(plain)
    (reg2 * 1024.0).backward()
    assert abs(float(reg) - float(reg2)) < 1e-6 * float(reg2)
    for a, p in zip(got, enc.planes_features_wavelet_coefs):
        np.testing.assert_allclose(a.cpu().numpy(), p.grad.cpu().numpy(), rtol=1e-6, atol=0)
        assert float(a.view(-1)[::7].abs().max()) == 0.0
    enc.zero_grad()
    enc.fused_l1_views = True
    v, p = enc.get_wavelet_features()[0], enc.planes_features_wavelet_coefs[0]
    for f in (lambda t: (t * 2).sum(), lambda t: torch.abs(t).sum(), lambda t: t.abs().mean(dim=0).sum(),
              lambda t: (t.abs() * 3 + 1).max(), lambda t: torch.mean(t.abs()), lambda t: t.abs().sum() / t.numel(),
              lambda t: (t ** 2).mean(), lambda t: t.abs()[0, 1].sum()):
        assert abs(float(f(v)) - float(f(p))) <= 1e-5 * abs(float(f(p))) + 1e-9
    with torch.no_grad():
        assert all(type(t) is torch.nn.Parameter for t in enc.get_wavelet_features())


def _tiny_encoder_model(cuda, seed):
    from trinerflet_amd.nerf.network import NeRFNetwork
    m = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, hidden_dim=64, hidden_dim_color=64,
                    triplane_channels=16, triplane_resolution=256, triplane_wavelet_levels=4).to(cuda)
    g = torch.Generator(device=cuda).manual_seed(seed)
    with torch.no_grad():
        for p in m.encoder.planes_features_wavelet_coefs:
            p.copy_(torch.randn(p.shape, generator=g, device=cuda) * 0.05)
            p.view(-1)[::11] = 0.0
    return m


def _reg_loss(enc, k, lam=0.3):
    """utils.py:639-655's regulariser plus a data term that reaches the same parameters through get_planes()."""
    enc.reset_cahce()
    planes = enc.get_planes()
    wf = enc.get_wavelet_features()
    tot = sum(v.numel() for v in wf)
    reg = sum(v.abs().mean() * (v.numel() / tot) for v in wf) / len(wf)
    return ((planes * (1.0 + 0.05 * k)) ** 2).mean() + lam * reg


def test_folded_regulariser_equals_the_materialised_gradient(cuda):
    """fold_l1: the backward of coef.abs().mean() leaves a scalar in the optimiser's sink, no gradient tensor; the pass adds
    s * sign(p) in registers.  Against torch.optim.Adam on the same loop under GradScaler, one iteration with a non-finite
    data gradient (skipped; the sink is emptied with it) and one with two backward passes before the step."""
    from trinerflet_amd.optim import FusedAdamL1
    ma, mb = _tiny_encoder_model(cuda, 3), _tiny_encoder_model(cuda, 3)
    mb.load_state_dict(ma.state_dict())
    ca, cb = list(ma.encoder.planes_features_wavelet_coefs), list(mb.encoder.planes_features_wavelet_coefs)
    pa, pb = ca + [ma.encoder.planes_features], cb + [mb.encoder.planes_features]
    oa = torch.optim.Adam(pa, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    ob = FusedAdamL1(pb, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    sa, sb = (torch.amp.GradScaler("cuda", init_scale=256.0, growth_interval=3) for _ in range(2))
    for k in range(8):
        for m, opt, sc in ((ma, oa, sa), (mb, ob, sb)):
            opt.zero_grad(set_to_none=True)
            for rep in range(2 if k == 6 else 1):
                loss = _reg_loss(m.encoder, k + rep)
                if k == 3:
                    loss = loss + m.encoder.planes_features.view(-1)[0] * float("inf")
                sc.scale(loss).backward()
            if opt is ob:
                # nothing of the regulariser was materialised: the coefficients' .grad is the data gradient alone
                assert all(s_.used for s_ in ob._sinks.values())
            sc.step(opt)
            sc.update()
        assert float(sa.get_scale()) == float(sb.get_scale()), k
        assert not any(s_.used for s_ in ob._sinks.values()) and float(next(iter(ob._sinks.values())).vec.abs().sum()) == 0.0
    for a, b in zip(pa, pb):
        # (g + s sign(p)) * inv against g * inv + (s * inv) sign(p): one rounding apart; where the data gradient all but
        # cancels the L1 term Adam's m / sqrt(v) magnifies that (21 of 590 k coefficients beyond 3e-5 relative), and a
        # coefficient that lands within a rounding of zero takes the other sign(p) on the next steps (one in 2.4 M; each such
        # step moves it by up to 2 lr = 2e-2 against the other run: seen 2e-3 .. 1.01e-2 over repeated runs)
        x, y = b.detach().cpu().numpy(), a.detach().cpu().numpy()
        bad = np.abs(x - y) > 3e-5 * np.abs(y) + 3e-7
        assert bad.mean() < 2e-4 and np.abs(x - y).max() < 4e-2, (bad.mean(), np.abs(x - y).max())
        assert float(oa.state[a]["step"]) == float(ob.state[b]["step"]) == 7.0
    # the fold is a property of the live optimiser: without it (or with fold_l1=False) the gradient is materialised
    ob.fold_l1 = False
    mb.load_state_dict(ma.state_dict())
    mb.encoder.zero_grad()
    _reg_loss(mb.encoder, 0).backward()
    ma.encoder.zero_grad()
    _reg_loss(ma.encoder, 0).backward()
    for a, b in zip(pa, pb):
        np.testing.assert_allclose(b.grad.cpu().numpy(), a.grad.cpu().numpy(), rtol=1e-5, atol=1e-9)
    # unscale_ before step is refused while a folded term is pending
    ob.fold_l1 = True
    ob.zero_grad()
    sb.scale(_reg_loss(mb.encoder, 0)).backward()
    sb.unscale_(ob)
    with pytest.raises(RuntimeError):
        sb.step(ob)


def test_parameters_frozen_by_clear_grad_are_skipped_like_torch_adam_skips_them(cuda):
    """Trainer.clear_grad() (nerf/utils.py:1105-1114, --min_wavelet_resolution_to_learn > 0) sets .grad = None after
    backward for the MLP, the LL band and the coarse wavelet levels.  torch.optim.Adam leaves such parameters, their
    moments and their `step` alone; with fold_l1 the regulariser's term sits in the sink, not in .grad, and must be dropped
    with it (round 4 stepped every fp32 parameter once any term was folded)."""
    from trinerflet_amd.optim import FusedAdamL1
    ma, mb = _tiny_encoder_model(cuda, 5), _tiny_encoder_model(cuda, 5)
    mb.load_state_dict(ma.state_dict())
    min_res = 64

    def clear_grad(model):
        with torch.no_grad():
            wavelet_grad = [v.grad for v in model.encoder.parameters()]
            for v in model.parameters():
                v.grad = None
            for i, v in enumerate(model.encoder.parameters()):
                if v.shape[-1] > min_res:
                    v.grad = wavelet_grad[i]
    oa = torch.optim.Adam(list(ma.parameters()), lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    ob = FusedAdamL1(list(mb.parameters()), lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    sa, sb = (torch.amp.GradScaler("cuda", init_scale=256.0) for _ in range(2))
    before = {k: v.detach().clone() for k, v in mb.named_parameters()}
    for k in range(4):
        for m, opt, sc in ((ma, oa, sa), (mb, ob, sb)):
            opt.zero_grad(set_to_none=True)
            if k == 2:       # one step with everything learnable in between: moments of the later-frozen parameters exist
                sc.scale(_reg_loss(m.encoder, k) + sum((p ** 2).sum() for p in m.sigma_net.parameters())).backward()
            else:
                sc.scale(_reg_loss(m.encoder, k)).backward()
                clear_grad(m)
            sc.step(opt)
            sc.update()
    frozen = learned = 0
    for (name, a), (_, b) in zip(ma.named_parameters(), mb.named_parameters()):
        sta, stb = oa.state.get(a, {}), ob.state.get(b, {})
        assert float(sta.get("step", 0.0)) == float(stb.get("step", 0.0)), name
        if float(sta.get("step", 0.0)) <= 1.0:
            frozen += 1
        else:
            learned += 1
            assert not torch.equal(b.detach(), before[name])
        x, y = b.detach().cpu().numpy(), a.detach().cpu().numpy()
        bad = np.abs(x - y) > 3e-5 * np.abs(y) + 3e-7
        assert bad.mean() < 2e-4, (name, bad.mean())
        if "exp_avg" in sta:
            ma_, mb_ = sta["exp_avg"].cpu().numpy(), stb["exp_avg"].cpu().numpy()     # (data gradients are float-atomic sums)
            np.testing.assert_allclose(mb_, ma_, rtol=1e-3, atol=1e-4 * float(np.abs(ma_).max()) + 1e-12)
    assert frozen >= 3 and learned >= 1
    assert not any(s_.used for s_ in ob._sinks.values())
    # opt-in: a parameter only the regulariser reached (no data gradient at all) is stepped
    oc = FusedAdamL1(list(mb.encoder.planes_features_wavelet_coefs), lr=1e-2, l1_without_grad=True)
    oc.zero_grad(set_to_none=True)
    wf = mb.encoder.get_wavelet_features()
    sum(v.abs().mean() for v in wf).backward()
    assert all(p.grad is None for p in mb.encoder.planes_features_wavelet_coefs)
    ref = [p.detach().clone() for p in mb.encoder.planes_features_wavelet_coefs]
    oc.step()
    assert all(not torch.equal(p.detach(), r) for p, r in zip(mb.encoder.planes_features_wavelet_coefs, ref))


def test_read_only_inf_check_and_both_gradscaler_hand_overs(cuda):
    """tnl_nonfinite_check at the edges of its partition, and FusedAdamL1.step driven through GradScaler's two contracts
    (the scaler passed as `grad_scaler`; the found_inf / grad_scale attributes) giving the same update."""
    from trinerflet_amd import _lib as L
    from trinerflet_amd.optim import FusedAdamL1
    n = 4 * 1024 * 4096 + 4 * 777 + 3
    x = torch.randn(n, device=cuda)
    found = torch.zeros(1, device=cuda)
    chk = lambda: L.check(L.lib().tnl_nonfinite_check(L.ptr(x), L.u64(n), L.ptr(found), L.stream()), "chk")
    chk()
    assert float(found) == 0.0
    for pos, val in ((0, float("inf")), (n - 1, float("nan")), (n - 4, float("-inf")), (4 * 1024 * 2000 + 5, float("nan")),
                     (4 * 1024 * 4096 + 17, float("inf"))):
        keep = float(x[pos])
        x[pos] = val
        found.zero_()
        chk()
        assert float(found) == 1.0, pos
        x[pos] = keep
    found.fill_(1.0)
    chk()
    assert float(found) == 1.0                     # never cleared
    runs = []
    for kwarg in (True, False):
        ps = _params(cuda, 5)
        opt = FusedAdamL1(_groups(ps), betas=(0.9, 0.99), eps=1e-15)
        sc = torch.amp.GradScaler("cuda", init_scale=512.0, growth_interval=2)
        for k in range(6):
            opt.zero_grad()
            loss = _loss(ps, k)
            if k == 2:
                loss = loss + ps[0].view(-1)[-1] * float("nan")
            sc.scale(loss).backward()
            if kwarg:
                sc.step(opt)
            else:                                   # what GradScaler.step does for an optimiser without the keyword
                state = sc._per_optimizer_states[id(opt)]
                sc._check_inf_per_device(opt)
                opt.grad_scale, opt.found_inf = sc._get_scale_async(), sum(state["found_inf_per_device"].values())
                opt.step()
                del opt.grad_scale, opt.found_inf
                state["stage"] = torch.amp.grad_scaler.OptState.STEPPED
            sc.update()
        runs.append(([p.detach().clone() for p in ps], float(sc.get_scale()), [float(opt.state[p]["step"]) for p in ps]))
    assert runs[0][1] == runs[1][1] and runs[0][2] == runs[1][2] == [5.0] * 5
    for a, b in zip(runs[0][0], runs[1][0]):
        assert torch.equal(a, b)


def test_patched_torch_adam_hands_out_the_fused_optimiser_where_it_can_stand_in(cuda):
    """install_dropin(fused_adam=True) -> optim.patch_torch_adam(): main_nerf.py:119's torch.optim.Adam(model.get_params(lr),
    betas=(0.9, 0.99), eps=1e-15) becomes a FusedAdamL1 (dense fp32 device parameters, plain options); anything else is
    torch's own Adam."""
    from trinerflet_amd import optim
    real = torch.optim.Adam
    try:
        optim.patch_torch_adam()
        ps = _params(cuda, 7)
        o = torch.optim.Adam(_groups(ps, 1e-2), lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
        assert type(o) is optim.FusedAdamL1 and isinstance(o, torch.optim.Optimizer)
        assert o.param_groups[1]["weight_decay"] == 1e-2 and o.param_groups[0]["betas"] == (0.9, 0.99)
        torch.optim.lr_scheduler.LambdaLR(o, lambda k: 0.5)                       # main_nerf.py:129 accepts it
        assert type(torch.optim.Adam(ps, amsgrad=True)) is real
        assert type(torch.optim.Adam(ps, fused=True)) is real
        assert type(torch.optim.Adam([torch.nn.Parameter(torch.zeros(3, dtype=torch.float64, device=cuda))])) is real
        assert type(torch.optim.Adam(ps + [torch.nn.Parameter(torch.zeros(3))])) is real          # a CPU parameter among them
    finally:
        optim.unpatch_torch_adam()
    assert torch.optim.Adam is real


def test_live_deferred_split_equals_the_whole_array_pass_bit_for_bit(cuda):
    """FusedAdamL1(defer=True): with the windowed rebuild under autograd only the live rectangle of a wavelet level is stepped
    every iteration, the rest replayed from a ring of step scalars (module docstring).  Against defer=False on the same
    loop under GradScaler -- with a skipped (non-finite) iteration, a window change, a whole-plane read in between (the
    density-grid refresh's get_planes_whole), more than 16 iterations in one period (ring full) and the folded regulariser
    -- parameters and both moments after a flush are THE SAME BITS, and so are the optimiser's state_dict()s."""
    from trinerflet_amd.nerf.network import NeRFNetwork
    from trinerflet_amd.optim import FusedAdamL1

    def make():
        m = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, hidden_dim=64, hidden_dim_color=64,
                        triplane_channels=16, triplane_resolution=512, triplane_wavelet_levels=8).to(cuda)
        g = torch.Generator(device=cuda).manual_seed(11)
        with torch.no_grad():
            for p in m.encoder.planes_features_wavelet_coefs:
                p.copy_(torch.randn(p.shape, generator=g, device=cuda) * 0.05)
        m.encoder.windowed_autograd = True
        return m
    ma, mb = make(), make()
    mb.load_state_dict(ma.state_dict())
    wins = [[128, 192, 64, 64, 128, 192, 256, 192], [64, 128, 192, 192, 64, 128, 256, 192]]
    cur = {"w": wins[0]}
    for m in (ma, mb):
        m.encoder.window_provider = lambda: list(cur["w"])
    pa, pb = list(ma.encoder.parameters()), list(mb.encoder.parameters())
    oa = FusedAdamL1(pa, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, defer=False)
    ob = FusedAdamL1(pb, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    sa, sb = (torch.amp.GradScaler("cuda", init_scale=256.0, growth_interval=5) for _ in range(2))

    def loss_of(m, k):
        enc = m.encoder
        enc.reset_cahce()
        planes = enc.get_planes()
        w = planes._tnl_window
        data = sum((planes[p, :, w[3 + p]:w[3 + p] + w[7], w[p]:w[p] + w[6]] * (1.0 + 0.03 * k)).pow(2).mean() for p in range(3))
        wf = enc.get_wavelet_features()
        tot = sum(v.numel() for v in wf)
        return data + 0.3 * sum(v.abs().mean() * (v.numel() / tot) for v in wf) / len(wf)
    lagged = False
    for k in range(44):
        if k == 21:
            cur["w"] = wins[1]
        for m, opt, sc in ((ma, oa, sa), (mb, ob, sb)):
            if k == 33:
                with torch.no_grad():
                    m.encoder.reset_cahce()
                    m.encoder.get_planes()            # whole planes outside autograd: every coefficient is read
            opt.zero_grad(set_to_none=True)
            loss = loss_of(m, k)
            if k == 7:
                loss = loss + m.encoder.planes_features.view(-1)[0] * float("inf")
            sc.scale(loss).backward()
            sc.step(opt)
            sc.update()
        assert float(sa.get_scale()) == float(sb.get_scale()), k
        if k == 12:      # mid-period: the deferred coefficients lag behind, the live ones do not
            fine_a, fine_b = pa[-1] if pa[-1].dim() == 5 else pa[-2], pb[-1] if pb[-1].dim() == 5 else pb[-2]
            lagged = not torch.equal(fine_a, fine_b)
            d = ob._deferred[fine_b]
            lv = d["live"]
            assert d["pending"] > 0
            for pl in range(3):
                sl = (pl, slice(None), slice(None), slice(lv[3 + pl], lv[3 + pl] + lv[7]), slice(lv[pl], lv[pl] + lv[6]))
                assert torch.equal(fine_a[sl], fine_b[sl])
    # (at R = 512 only the finest level's live rectangle is small enough to split: one deferred step per iteration)
    assert lagged and ob.deferred_steps >= 43 and ob.deferred_flushes >= 3 and oa.deferred_steps == 0      # ring full, window change, whole-plane read
    sda, sdb = oa.state_dict(), ob.state_dict()            # (flushes)
    assert all(d["pending"] == 0 for d in ob._deferred.values())
    for a, b in zip(pa, pb):
        assert torch.equal(a, b), a.shape
        for key in ("exp_avg", "exp_avg_sq", "step"):
            assert torch.equal(oa.state[a][key], ob.state[b][key]), (a.shape, key)
    for ka in sda["state"]:
        for key in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(sda["state"][ka][key], sdb["state"][ka][key])
    # the encoder's own state_dict flushes as well
    ob.zero_grad(set_to_none=True)
    sb.scale(loss_of(mb, 50)).backward()
    sb.step(ob)
    assert any(d["pending"] for d in ob._deferred.values())
    mb.state_dict()
    assert all(d["pending"] == 0 for d in ob._deferred.values())


@pytest.mark.parametrize("order", ["reg_first", "reg_last", "extra_term"])
def test_a_dense_gradient_on_top_of_the_windowed_chain_is_not_split(cuda, order):
    """the live / deferred split must be taken only while .grad is the windowed chain's own, untouched tensor.
    With fold_l1=False the regulariser's gradient (utils.py:639-655) is materialised and reaches every coefficient; the
    engine sums it with the chain's gradient in the leaf's input buffer -- in place, possibly IN the chain's tensor, whose
    address then still matches what the backward recorded.  The version counter recorded beside the address catches it: the
    step takes the whole-array pass, and defer=True leaves the same bits as defer=False.  (Before the fix the split was
    taken and the L1 pull outside the rectangle silently dropped: g = 0 replayed there.)"""
    from trinerflet_amd.nerf.network import NeRFNetwork
    from trinerflet_amd.optim import FusedAdamL1

    def make():
        m = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, hidden_dim=64, hidden_dim_color=64,
                        triplane_channels=16, triplane_resolution=512, triplane_wavelet_levels=8).to(cuda)
        g = torch.Generator(device=cuda).manual_seed(13)
        with torch.no_grad():
            for p in m.encoder.planes_features_wavelet_coefs:
                p.copy_(torch.randn(p.shape, generator=g, device=cuda) * 0.05)
        m.encoder.windowed_autograd = True
        m.encoder.window_provider = lambda: [128, 192, 64, 64, 128, 192, 256, 192]
        return m
    ma, mb = make(), make()
    mb.load_state_dict(ma.state_dict())
    pa, pb = list(ma.encoder.parameters()), list(mb.encoder.parameters())
    fold = order == "extra_term"        # (the extra term is dense by itself: the regulariser may stay folded there)
    oa = FusedAdamL1(pa, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, defer=False, fold_l1=fold)
    ob = FusedAdamL1(pb, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, defer=True, fold_l1=fold)

    def loss_of(m, k):
        enc = m.encoder
        enc.reset_cahce()

        def reg():
            wf = enc.get_wavelet_features()
            tot = sum(v.numel() for v in wf)
            return 0.3 * sum(v.abs().mean() * (v.numel() / tot) for v in wf) / len(wf)

        def data():
            planes = enc.get_planes()
            w = planes._tnl_window
            return sum((planes[p, :, w[3 + p]:w[3 + p] + w[7], w[p]:w[p] + w[6]] * (1.0 + 0.03 * k)).pow(2).mean()
                       for p in range(3))
        if order == "reg_first":
            r = reg()
            return r + data()
        if order == "reg_last":
            d = data()
            return d + reg()
        # another loss term that touches the coefficients directly, recorded before the render
        extra = sum(1e-3 * (q * q).mean() for q in enc.planes_features_wavelet_coefs)
        return extra + data() + reg()
    for k in range(6):
        for m, opt in ((ma, oa), (mb, ob)):
            opt.zero_grad(set_to_none=True)
            loss_of(m, k).backward()
            opt.step()
    assert ob.deferred_steps == 0, ob.deferred_steps          # every gradient was dense: no step may take the split
    ob.flush_deferred()
    for a, b in zip(pa, pb):
        assert torch.equal(a, b), a.shape
        for key in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(oa.state[a][key], ob.state[b][key]), (a.shape, key)
    # the pull did reach the coefficients outside the live rectangle (far corner of the finest level)
    fine = [q for q in pb if q.dim() == 5][-1]
    assert float(ob.state[fine]["exp_avg"][0, 0, 0, :8, :8].abs().max()) > 0


class _TorchEmaLike:
    """torch_ema.ExponentialMovingAverage's surface as the reference's Trainer uses it (utils.py:516-520, 838-841, 890,
    1204-1207): shadow copies updated from the parameter tensors, store / copy_to / restore around an evaluation."""

    def __init__(self, parameters, decay):
        self.params = list(parameters)
        self.decay = decay
        self.shadow = [p.detach().clone() for p in self.params]
        self.collected = None

    def update(self):
        with torch.no_grad():
            for s, p in zip(self.shadow, self.params):
                s.sub_((1.0 - self.decay) * (s - p))

    def store(self):
        self.collected = [p.detach().clone() for p in self.params]

    def copy_to(self):
        with torch.no_grad():
            for s, p in zip(self.shadow, self.params):
                p.copy_(s)

    def restore(self):
        with torch.no_grad():
            for c, p in zip(self.collected, self.params):
                p.copy_(c)


def test_reference_loop_ema_beside_the_deferred_split(cuda, monkeypatch):
    """the reference's loop keeps a torch_ema EMA (run_utils.py:93 default 0.95): update() after every epoch,
    store() / copy_to() / restore() around every evaluation -- all of them read or write the parameter tensors directly.
    FusedAdamL1(defer=True) guards the loaded torch_ema class (optim.guard_ema_class): each of those calls first replays the
    pending steps.  Against defer=False: parameters, moments AND the EMA's shadow values are the same bits after two
    "epochs" of 10 iterations (pending steps at each EMA call) with an evaluation's whole-plane read in between."""
    import sys
    import types
    from trinerflet_amd import optim
    from trinerflet_amd.nerf.network import NeRFNetwork
    fake = types.ModuleType("torch_ema")
    fake.ExponentialMovingAverage = type("ExponentialMovingAverage", (_TorchEmaLike,), {})      # (a fresh class per test)
    monkeypatch.setitem(sys.modules, "torch_ema", fake)

    def make():
        m = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, hidden_dim=64, hidden_dim_color=64,
                        triplane_channels=16, triplane_resolution=512, triplane_wavelet_levels=8).to(cuda)
        g = torch.Generator(device=cuda).manual_seed(17)
        with torch.no_grad():
            for p in m.encoder.planes_features_wavelet_coefs:
                p.copy_(torch.randn(p.shape, generator=g, device=cuda) * 0.05)
        m.encoder.windowed_autograd = True
        m.encoder.window_provider = lambda: [128, 192, 64, 64, 128, 192, 256, 192]
        return m
    ma, mb = make(), make()
    mb.load_state_dict(ma.state_dict())
    pa, pb = list(ma.encoder.parameters()), list(mb.encoder.parameters())
    oa = optim.FusedAdamL1(pa, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, defer=False)
    ob = optim.FusedAdamL1(pb, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, defer=True)
    assert fake.ExponentialMovingAverage._tnl_flush_guard           # guarded when the optimiser was made
    ea, eb = fake.ExponentialMovingAverage(pa, 0.95), fake.ExponentialMovingAverage(pb, 0.95)

    def loss_of(m, k):
        enc = m.encoder
        enc.reset_cahce()
        planes = enc.get_planes()
        w = planes._tnl_window
        data = sum((planes[p, :, w[3 + p]:w[3 + p] + w[7], w[p]:w[p] + w[6]] * (1.0 + 0.03 * k)).pow(2).mean() for p in range(3))
        wf = enc.get_wavelet_features()
        tot = sum(v.numel() for v in wf)
        return data + 0.3 * sum(v.abs().mean() * (v.numel() / tot) for v in wf) / len(wf)
    evals = []
    for epoch in range(2):
        for k in range(10):
            for m, opt in ((ma, oa), (mb, ob)):
                opt.zero_grad(set_to_none=True)
                loss_of(m, 10 * epoch + k).backward()
                opt.step()
        assert any(d["pending"] for d in ob._deferred.values())         # the EMA calls below meet pending steps
        for m, ema in ((ma, ea), (mb, eb)):
            ema.update()                                                 # utils.py:1204-1207
            ema.store()                                                  # utils.py:838-841: evaluate with the averaged weights
            ema.copy_to()
            with torch.no_grad():
                m.encoder.reset_cahce()
                evals.append(m.encoder.get_planes().float().clone())
            ema.restore()                                                # :890
            m.encoder.reset_cahce()
        assert torch.equal(evals[-1], evals[-2])
    assert ob.deferred_steps >= 18 and oa.deferred_steps == 0
    ob.flush_deferred()
    for a, b, sa_, sb_ in zip(pa, pb, ea.shadow, eb.shadow):
        assert torch.equal(a, b) and torch.equal(sa_, sb_), a.shape
        for key in ("exp_avg", "exp_avg_sq", "step"):
            assert torch.equal(oa.state[a][key], ob.state[b][key]), (a.shape, key)


def test_model_load_state_dict_meets_no_pending_steps(cuda):
    """model.load_state_dict() while steps are pending -- the reference loads the model, then the optimiser
    (utils.py:1482-1510).  The encoder's load_state_dict pre-hook replays the pending steps BEFORE the copy, so nothing of
    the old run is replayed on top of the loaded values afterwards."""
    from trinerflet_amd.nerf.network import NeRFNetwork
    from trinerflet_amd.optim import FusedAdamL1
    m = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, hidden_dim=64, hidden_dim_color=64,
                    triplane_channels=16, triplane_resolution=512, triplane_wavelet_levels=8).to(cuda)
    g = torch.Generator(device=cuda).manual_seed(19)
    with torch.no_grad():
        for p in m.encoder.planes_features_wavelet_coefs:
            p.copy_(torch.randn(p.shape, generator=g, device=cuda) * 0.05)
    ckpt = {k: v.clone() for k, v in m.state_dict().items()}
    m.encoder.windowed_autograd = True
    m.encoder.window_provider = lambda: [128, 192, 64, 64, 128, 192, 256, 192]
    ps = list(m.encoder.parameters())
    opt = FusedAdamL1(ps, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    for k in range(5):
        opt.zero_grad(set_to_none=True)
        m.encoder.reset_cahce()
        planes = m.encoder.get_planes()
        w = planes._tnl_window
        loss = sum(planes[p, :, w[3 + p]:w[3 + p] + w[7], w[p]:w[p] + w[6]].pow(2).mean() for p in range(3))
        wf = m.encoder.get_wavelet_features()
        (loss + 0.3 * sum(v.abs().mean() for v in wf) / len(wf)).backward()
        opt.step()
    assert any(d["pending"] for d in opt._deferred.values())
    m.load_state_dict(ckpt)
    assert all(d["pending"] == 0 for d in opt._deferred.values())
    opt.flush_deferred()
    for k, v in m.state_dict().items():
        assert torch.equal(v, ckpt[k]), k
