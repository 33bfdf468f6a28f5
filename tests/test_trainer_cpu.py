"""Host-side pieces of the trainer row against vectors produced by the reference's own
functions (tests/golden/make_golden_trainer.py): LR schedule, PSNR meter, ray generation oracle, the pixel
permutation, checkpoint component layouts.  No GPU."""
import os

import numpy as np
import torch

from oracle import cref


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "trainer_reference.npz"))


def test_lr_schedule_matches_reference_decay_function(golden_dir):
    from trinerflet_amd.train import lr_factor
    from oracle.field import lr_factor as lr_oracle
    g = _g(golden_dir)
    for it, (iters, warm), want in zip(g["lr/it"], g["lr/cfg"], g["lr/factor"]):
        assert abs(lr_factor(int(it), int(iters), int(warm)) - want) <= 1e-12 * max(1.0, abs(want))
        assert abs(lr_oracle(int(it), int(iters), int(warm)) - want) <= 1e-12 * max(1.0, abs(want))


def test_psnr_meter_matches_reference(golden_dir):
    from trinerflet_amd.trainer import PSNRMeter
    g = _g(golden_dir)
    m = PSNRMeter()
    for k in range(2):
        m.update(torch.from_numpy(g["psnr/pred"][k:k + 1]), torch.from_numpy(g["psnr/truth"][k:k + 1]))
    assert abs(m.measure() - float(g["psnr/value"])) < 1e-5


def test_oracle_get_rays_matches_reference(golden_dir):
    g = _g(golden_dir)
    H, W = (int(v) for v in g["rays/HW"])
    B = g["rays/poses"].shape[0]
    assert np.array_equal(g["rays/inds"], np.tile(np.arange(H * W), (B, 1)))     # raster order, utils.py:135-136
    o, d = cref.get_rays(g["rays/poses"], g["rays/intrinsics"], H, W, np.arange(B * H * W))
    assert np.array_equal(o, g["rays/rays_o"].reshape(-1, 3))
    np.testing.assert_allclose(d, g["rays/rays_d"].reshape(-1, 3), rtol=0, atol=2e-7)


def test_batch_contract_of_the_reference(golden_dir):
    """shuffle_data flattens [B,N,..] to B*N rows in a random order; select_batch slices num_rays rows (short last
    batch) and adds a leading 1 -- what RayPool.batch reproduces (without the leading axis)."""
    g = _g(golden_dir)
    assert list(g["batch/shuffled_rows"]) == [10, 3] and bool(g["batch/is_perm"])
    assert list(g["batch/sel_shape"]) == [1, 4, 3] and list(g["batch/last_shape"]) == [1, 2, 3]


def test_permutation_is_a_bijection_and_library_agrees():
    from trinerflet_amd import _lib as L
    lib = L.lib()          # host function of the shared library: callable without a GPU
    for total, key in [(1, 5), (2, 5), (7, 1), (1000, 2), (4097, 0x1234567), (65536, 9), (100000, 77)]:
        p = cref.permute_index(np.arange(total), total, key)
        assert np.array_equal(np.sort(p), np.arange(total)), total
        for g_ in (0, total // 3, total - 1):
            assert lib.tnl_permute_index(L.u64(g_), L.u64(total), L.u64(key)) == int(p[g_])
    a = cref.permute_index(np.arange(4097), 4097, 1)
    b = cref.permute_index(np.arange(4097), 4097, 2)
    assert (a == b).mean() < 0.01 and abs(np.corrcoef(a, np.arange(4097))[0, 1]) < 0.1


def test_sample_pdf_matches_reference(golden_dir):
    """Importance resampling of NeRFRenderer.run (renderer.py:18-55), deterministic mode, incl. an empty ray."""
    from trinerflet_amd.nerf.renderer import sample_pdf
    g = _g(golden_dir)
    got = sample_pdf(torch.from_numpy(g["pdf/bins"]), torch.from_numpy(g["pdf/weights"]), 16, det=True)
    np.testing.assert_allclose(got.numpy(), g["pdf/samples"], rtol=1e-6, atol=1e-6)
