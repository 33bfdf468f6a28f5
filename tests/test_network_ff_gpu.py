"""The `--ff` module API (reconstruction/nerf/network_ff.py; aux_libs/ffmlp/ffmlp.py) on this build: FFMLP's arithmetic
contract (fp16 operands, ReLU, padded output layer, first output_dim columns) against an fp64 restatement, and the network
through the renderer's training and inference branches under autograd."""
import numpy as np
import pytest
import torch

from trinerflet_amd import synthetic


def test_ffmlp_against_its_definition_on_the_host():
    from trinerflet_amd.ffmlp import FFMLP
    for args in ((48, 16, 64, 2), (32, 3, 64, 3)):
        m = FFMLP(*args)
        x = torch.randn(257, args[0], generator=torch.Generator().manual_seed(1))       # (any batch size: no 128 padding needed)
        y = m(x)
        assert y.shape == (257, args[1]) and y.dtype == torch.float16
        mats = [w.detach().half().double() for w in m.matrices()]
        assert [tuple(w.shape) for w in mats] == [(args[2], args[0])] + [(args[2], args[2])] * (args[3] - 1) + [(16, args[2])]
        h = x.half().double()
        for W in mats[:-1]:
            h = torch.relu(h @ W.T).half().double()          # fp16 activations between the layers
        ref = (h @ mats[-1].T)[:, :args[1]]
        assert float((y.double() - ref).abs().max()) < 2e-2 * float(ref.abs().max())
        y.float().pow(2).sum().backward()
        assert m.weights.grad is not None and float(m.weights.grad.abs().sum()) > 0
        # the padded rows of the output layer take no gradient
        pad = m.matrices(m.weights.grad)[-1][args[1]:]
        assert pad.numel() == 0 or float(pad.abs().max()) == 0.0


@pytest.mark.gpu
def test_ff_network_renders_and_trains_through_autograd(cuda):
    from trinerflet_amd.nerf.network_ff import NeRFNetwork
    m = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, density_thresh=10, hidden_dim=64,
                    hidden_dim_color=64, triplane_channels=16, triplane_resolution=128, triplane_wavelet_levels=2,
                    wavelet_type="bior6.8").to(cuda)
    assert not m._fused_ok()
    with torch.no_grad():
        m.encoder.planes_features.normal_(0, 0.3)
        for p in m.encoder.planes_features_wavelet_coefs:
            p.normal_(0, 0.05)
    m.density_bitfield.copy_(torch.from_numpy(synthetic.sphere_bitfield(128, 2, 1.5, 0.8, 0.0)).to(cuda))
    o, d = synthetic.training_rays(512, n_cams=4, seed=7)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    # forward == its composition (network_ff.py:52-75)
    x = (torch.rand(1000, 3, device=cuda) * 2 - 1) * 1.4
    dirs = torch.nn.functional.normalize(torch.randn(1000, 3, device=cuda), dim=-1)
    with torch.no_grad():
        sigma, rgb = m(x, dirs)
        h = m.sigma_net(m.encoder(x, bound=m.bound))
        assert torch.equal(sigma, torch.exp(h[..., 0].float())) and sigma.dtype == torch.float32
        dd = m.encoder_dir(dirs)
        hc = torch.cat([dd.to(h.dtype), h[..., 1:], torch.zeros_like(h[..., :1])], -1)
        assert torch.equal(rgb, torch.sigmoid(m.color_net(hc))) and rgb.shape == (1000, 3)
        dens = m.density(x)
        assert torch.equal(dens["sigma"], sigma) and dens["geo_feat"].shape == (1000, 15)
        mask = torch.rand(1000, device=cuda) > 0.5
        col = m.color(x, dirs, mask=mask, geo_feat=dens["geo_feat"])
        assert torch.equal(col[mask].to(rgb.dtype), rgb[mask]) and float(col[~mask].abs().max()) == 0.0
    # the training branch of run_cuda under autograd + an optimiser step (the reference Trainer's loop)
    m.train()
    opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    gt = t(synthetic.target_colors(d))
    m.mean_count = 0
    losses = []
    for _ in range(4):
        opt.zero_grad(set_to_none=True)
        m.encoder.reset_cahce(); m.encoder.get_planes()
        out = m.render(t(o)[None], t(d)[None], staged=False, bg_color=0, perturb=True, force_all_rays=True, dt_gamma=0,
                       max_steps=1024)
        loss = ((out["image"][0] - gt) ** 2).mean()
        loss.backward()
        for name in ("sigma_net.weights", "color_net.weights", "encoder.planes_features"):
            g = dict(m.named_parameters())[name].grad
            assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().sum()) > 0, name
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0]
    # the inference branch (the alive-ray loop: this architecture is outside the one-kernel render) and the grid refresh
    m.eval()
    with torch.no_grad():
        img = m.render(t(o)[None], t(d)[None], staged=True, bg_color=1.0, perturb=False, max_steps=1024)
        assert img["image"].shape == (1, 512, 3) and bool(torch.isfinite(img["image"]).all())
        m.update_extra_state()
    assert int(m.density_bitfield.count_nonzero()) >= 0 and np.isfinite(m.mean_density)
