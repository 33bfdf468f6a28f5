"""Degenerate inputs through the hot path: nothing occupied, nothing sampled, empty batches, one ray.  The reference
has no tests of its own; these are the cases its kernels guard with `if (n >= N) return` and its
wrappers with zero-filled buffers (raymarching.py:205-207,283-284)."""
import numpy as np
import pytest
import torch

from trinerflet_amd import synthetic

pytestmark = pytest.mark.gpu


def _model(dev, R=128, levels=2):
    from trinerflet_amd.nerf.network import NeRFNetwork
    torch.manual_seed(0)
    m = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, density_thresh=10, hidden_dim=64,
                    hidden_dim_color=64, triplane_channels=16, triplane_resolution=R, triplane_wavelet_levels=levels,
                    wavelet_type="bior6.8").to(dev)
    synthetic.init_field_parameters(m, seed=1)
    return m


def test_step_with_empty_occupancy_and_with_one_ray(cuda):
    """No occupied cell -> no sample: the step still runs (loss = background error + regulariser), the field weights
    see zero gradient, the coefficients move only through the L1 term; then a single-ray batch."""
    from trinerflet_amd.train import TrainStep
    m = _model(cuda)
    m.density_bitfield.zero_()
    ts = TrainStep(m, update_extra_interval=0, wavelet_regularization=0.1)
    m.mean_count = 0
    o, d = synthetic.training_rays(512, n_cams=3, seed=1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    gt = t(synthetic.target_colors(d))
    w_before = [w.detach().clone() for w in ts.Ws]
    for _ in range(2):                           # second call takes the occupancy-window path (empty window)
        loss = ts.step(t(o), t(d), gt)
    assert int(ts.last["counter"][0]) == 0 and ts.last["M"] >= 0
    assert torch.isfinite(loss) and abs(float(ts.last["mse"]) - float((gt ** 2).mean())) < 1e-6
    assert float(ts.last["found_inf"]) == 0.0
    for a, b in zip(w_before, ts.Ws):
        assert torch.equal(a, b.detach())        # Adam with g = 0 and m = v = 0 leaves the weights where they are
    # one ray through an occupied scene
    m2 = _model(cuda)
    m2.density_bitfield.copy_(t(synthetic.sphere_bitfield(128, m2.cascade, 1.5, 0.8, 0.0)))
    ts2 = TrainStep(m2, update_extra_interval=0)
    m2.mean_count = 0
    hit = int(np.argmin(np.linalg.norm(np.cross(o, d), axis=1)))      # the ray passing closest to the origin
    loss = ts2.step(t(o[hit:hit + 1]), t(d[hit:hit + 1]), gt[hit:hit + 1])
    assert torch.isfinite(loss) and int(ts2.last["counter"][0]) > 100


def test_empty_batches_through_the_c_abi(cuda):
    from trinerflet_amd import raymarching
    from trinerflet_amd.nerf import field as F_
    from trinerflet_amd.raypool import RayPool
    # zero samples: sort + reduce write an all-zero gradient
    C, R = 16, 64
    xyz = torch.empty(0, 3, device=cuda)
    dfeat = torch.empty(3, 0, C, dtype=torch.float16, device=cuda)
    g = torch.full((3, C, R, R), 9.0, device=cuda)
    F_.plane_grad_binned(dfeat, xyz, 1.0, C, R, g, channel_major=True)
    assert float(g.abs().sum()) == 0
    # m_actual = 0 on a non-empty buffer: nothing is read or accumulated
    xyz = torch.rand(1000, 3, device=cuda) - 0.5
    dfeat = torch.randn(3, 1000, C, device=cuda).half()
    zero = torch.zeros(1, dtype=torch.int32, device=cuda)
    g.fill_(9.0)
    F_.plane_grad_binned(dfeat, xyz, 1.0, C, R, g, m_actual=zero, channel_major=True)
    assert float(g.abs().sum()) == 0
    # zero rays
    e3 = torch.empty(0, 3, device=cuda)
    nears, fars = raymarching.near_far_from_aabb(e3, e3, torch.tensor([-1., -1, -1, 1, 1, 1], device=cuda), 0.2)
    assert nears.shape == (0,) and fars.shape == (0,)
    ws, depth, image = raymarching.composite_rays_train(torch.empty(0, device=cuda), torch.empty(0, 3, device=cuda),
                                                        torch.empty(0, 2, device=cuda),
                                                        torch.empty(0, 3, dtype=torch.int32, device=cuda))
    assert ws.shape == (0,) and image.shape == (0, 3)
    # a pool whose last batch holds a single pixel
    poses, intr, images = synthetic.sphere_dataset(1, 5, 5, seed=0)
    pool = RayPool(poses, intr, 5, 5, images, device=cuda)
    pool.shuffle(3)
    assert pool.steps_per_epoch(8) == 4
    last = pool.batch(3, 8)
    assert last["rays_o"].shape == (1, 3) and last["gt_rgb"].shape == (1, 3)
    with pytest.raises(IndexError):
        pool.batch(4, 8)


def test_inference_of_rays_that_miss_everything(cuda):
    m = _model(cuda)
    m.density_bitfield.zero_()
    m.eval()
    o, d = synthetic.training_rays(300, n_cams=2, seed=2)
    o, d = torch.from_numpy(o).to(cuda)[None], torch.from_numpy(d).to(cuda)[None]
    with torch.no_grad():
        for dev_loop in (True, False):
            out = m.render(o, d, staged=True, bg_color=0.25, perturb=False, max_steps=64, device_loop=dev_loop)
            assert float(out["weights_sum"].abs().max()) == 0
            assert torch.allclose(out["image"], torch.full_like(out["image"], 0.25))
