"""The module path (reconstruction/nerf/renderer.py:196-245's training branch on the drop-in modules, through autograd) with
its two savings -- the layout pass restricted to the occupancy window, the sample budget's zero padding skipped in the
fused field -- against the same path without them: same image, same gradients."""
import numpy as np
import pytest
import torch

from trinerflet_amd import synthetic

pytestmark = pytest.mark.gpu


def _model(cuda, R=512, C=16, fp16=True):
    from trinerflet_amd.nerf.network import NeRFNetwork
    m = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, density_scale=1, min_near=0.2,
                    density_thresh=10, bg_radius=-1, hidden_dim=64, hidden_dim_color=64, triplane_channels=C,
                    triplane_resolution=R, triplane_wavelet_levels=8, wavelet_type="bior6.8",
                    plane_dtype=torch.float16 if fp16 else torch.float32).to(cuda)
    m.encoder.windowed_autograd = True                # what install_dropin() turns on (TriPlaneVolume._autograd_window)
    synthetic.init_field_parameters(m, seed=0)
    m.density_bitfield.copy_(torch.from_numpy(synthetic.sphere_bitfield(128, m.cascade, 1.5, 0.45, 0.0)).to(cuda))
    m.train()
    return m


def _step(m, o, d, nz, mean_count):
    m.mean_count = mean_count
    m.zero_grad(set_to_none=True)
    m.encoder.reset_cahce()
    m.encoder.get_planes()
    out = m.render(o[None], d[None], staged=False, bg_color=0.0, perturb=True, force_all_rays=False, noises=nz, dt_gamma=0,
                   max_steps=256)
    (out["image"][0] ** 2).mean().backward()
    grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    return out["image"][0].detach().clone(), grads, int(m.step_counter[(m.local_step - 1) % 16][0])


def test_window_of_the_layout_pass_and_skipped_padding_change_nothing(cuda):
    m = _model(cuda)
    o, d = synthetic.training_rays(4096, n_cams=12, seed=1, H=100, W=100)
    o, d = torch.from_numpy(o).to(cuda), torch.from_numpy(d).to(cuda)
    nz = torch.rand(o.shape[0], device=cuda, generator=torch.Generator(device=cuda).manual_seed(2))
    _, _, count = _step(m, o, d, nz, 0)
    assert count > 10000
    budget = int(count * 1.25)                      # 20 % of the rows are padding at the origin
    win = m._occupancy_window()
    assert win is not None and win[6] * win[7] < 0.5 * 512 * 512
    img1, g1, c1 = _step(m, o, d, nz, budget)
    # whole-plane layout pass: nothing the forward reads differs -> the same image to the bit; the backward does not read
    # the copy at all (its tile lists are filled through atomics: the sums are repeatable to rounding only)
    m.use_occupancy_window = False
    img2, g2, c2 = _step(m, o, d, nz, budget)
    assert c2 == c1 == count and torch.equal(img2, img1) and g2.keys() == g1.keys()
    for n in g1:
        a, b = g2[n].cpu().numpy(), g1[n].cpu().numpy()
        assert np.abs(a - b).max() <= 2e-5 * (np.abs(a).max() + 1e-30), n
    # and with the padding evaluated like any other row (zero upstream gradient, all on the origin's texel): the same
    # image; the gradients agree up to the order of the sums in the one tile the padding rows are sorted into
    img0, g0, c0 = _step_plain(m, o, d, nz, budget)
    assert c0 == count and torch.equal(img0, img1) and g0.keys() == g1.keys()
    for n in g0:
        a, b = g0[n].cpu().numpy(), g1[n].cpu().numpy()
        scale = np.abs(a).max() + 1e-30
        assert np.abs(a - b).max() <= 2e-5 * scale, (n, np.abs(a - b).max() / scale)


def _step_plain(m, o, d, nz, mean_count):
    """_step with the march's counter withheld from the field (rows past it are computed) and whole-plane layout."""
    cls = type(m)
    orig = cls.forward

    def fwd(self, x, dd):
        keep, self._march_count = self._march_count, None
        try:
            return orig(self, x, dd)
        finally:
            self._march_count = keep
    cls.forward = fwd
    try:
        return _step(m, o, d, nz, mean_count)
    finally:
        cls.forward = orig


def test_partial_copy_is_replaced_when_more_is_asked_for(cuda):
    """The cached sampler copy made for a window is not served to a reader of the whole planes (the density-grid refresh,
    an evaluation render)."""
    m = _model(cuda, R=256, C=16)
    enc = m.encoder
    enc.reset_cahce()
    with torch.no_grad():
        whole = enc.get_planes_texel_major().clone()
        enc.reset_cahce()
        win = (64, 0, 128, 128, 64, 0, 128, 64)
        part = enc.get_planes_texel_major(window=win)
        assert enc._planes_tm_window == win
        for p in range(3):
            sl = (slice(win[3 + p], win[3 + p] + win[7]), slice(win[p], win[p] + win[6]))
            assert torch.equal(part[p][sl], whole[p][sl])
        assert enc.get_planes_texel_major(window=win) is part
        again = enc.get_planes_texel_major()
        assert again is not part and enc._planes_tm_window is None and torch.equal(again, whole)
        assert enc.get_planes_texel_major(window=win) is again            # a whole copy serves every window


def test_windowed_rebuild_under_autograd_and_the_refresh_sequence(cuda):
    """get_planes() of a training iteration builds only the window (and says so on the tensor); the density-grid refresh that
    follows it in the reference's loop (utils.py:1138-1146) needs every texel: the cached planes are replaced by whole,
    still differentiable ones, and the iteration's gradients reach the parameters."""
    m = _model(cuda)
    enc = m.encoder
    o, d = synthetic.training_rays(2048, n_cams=12, seed=3, H=100, W=100)
    o, d = torch.from_numpy(o).to(cuda), torch.from_numpy(d).to(cuda)
    nz = torch.rand(o.shape[0], device=cuda, generator=torch.Generator(device=cuda).manual_seed(4))
    enc.reset_cahce()
    planes = enc.get_planes()
    win = m._occupancy_window()
    assert planes._tnl_window == tuple(win) and planes.requires_grad
    with torch.no_grad():
        enc.reset_cahce()
        whole = enc.get_planes().clone()
        assert getattr(enc.get_planes(), "_tnl_window", None) is None           # no_grad: always whole
    enc.reset_cahce()
    planes = enc.get_planes()
    for p in range(3):
        sl = (slice(win[3 + p], win[3 + p] + win[7]), slice(win[p], win[p] + win[6]))
        assert torch.equal(planes[p][:, sl[0], sl[1]], whole[p][:, sl[0], sl[1]])
    # the refresh: density queries all over the volume
    m.update_extra_state()
    after = enc.get_planes()
    assert getattr(after, "_tnl_window", None) is None and after.requires_grad and torch.equal(after.detach(), whole)
    m.mean_count = 0
    m.zero_grad(set_to_none=True)
    out = m.render(o[None], d[None], staged=False, bg_color=0.0, perturb=True, force_all_rays=False, noises=nz, dt_gamma=0,
                   max_steps=256)
    (out["image"][0] ** 2).mean().backward()
    assert all(p.grad is not None and float(p.grad.abs().max()) > 0 for p in enc.planes_features_wavelet_coefs)
    # a field query outside run_cuda while only a window exists: whole planes are built first
    m.density_bitfield.copy_(torch.from_numpy(synthetic.sphere_bitfield(128, m.cascade, 1.5, 0.45, 0.0)).to(cuda))
    enc.reset_cahce()
    assert enc.get_planes()._tnl_window == tuple(win)
    x = (torch.rand(4096, 3, device=cuda) * 2 - 1) * 1.4                    # all over the volume
    dd = torch.nn.functional.normalize(torch.randn(4096, 3, device=cuda), dim=-1)
    s1, c1 = m(x, dd)
    assert getattr(enc.get_planes(), "_tnl_window", None) is None
    with torch.no_grad():
        enc.reset_cahce()
        s0, c0 = m(x, dd)
    assert torch.equal(s1.detach(), s0) and torch.equal(c1.detach(), c0)


def test_windowed_rebuild_is_opt_in_and_readers_outside_autograd_get_every_texel(cuda):
    from trinerflet_amd.triplaneencoder import triplane_encoder as te
    assert te.WINDOWED_AUTOGRAD is False                              # the library default (install_dropin() turns it on)
    m = _model(cuda, R=256)
    enc = m.encoder
    enc.windowed_autograd = False
    enc.reset_cahce()
    assert getattr(enc.get_planes(), "_tnl_window", None) is None      # default: whole planes, whatever the density grid
    with torch.no_grad():
        whole = enc.get_planes().clone()
    enc.windowed_autograd = True
    enc.reset_cahce()
    part = enc.get_planes()
    assert part._tnl_window is not None and enc.get_planes() is part  # under autograd the windowed cache is served
    with torch.no_grad():                                              # save_triplane, evaluation, ...
        seen = enc.get_planes()
    assert getattr(seen, "_tnl_window", None) is None and torch.equal(seen.detach(), whole) and seen.requires_grad
