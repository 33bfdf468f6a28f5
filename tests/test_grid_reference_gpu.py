"""A12 on the GPU: NeRFRenderer.mark_untrained_grid / update_extra_state of this build (restructured: Morton-ordered
cell table, one read-back, running-count picks, device-side threshold) against the outputs of the REFERENCE's own
methods (tests/golden/grid_reference.npz, see tests/golden/make_golden_grid.py and tests/test_grid_pins.py), with the
same analytic density and the same seeded draws handed in through `draws=`.  Grids bit for bit, mean_count exactly,
bitfields bit for bit except cells whose density sits within 1e-5 (relative) of the threshold when the threshold is
the GPU's own fp32 mean."""
import os

import numpy as np
import pytest
import torch

from oracle import cref, grid as ogrid
from tests.test_grid_pins import STEPS, blobs_of, check_grid

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref(golden_dir):
    return np.load(os.path.join(golden_dir, "grid_reference.npz"))


def _model(dev, H, thresh):
    from trinerflet_amd.nerf.network import NeRFNetwork
    m = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, density_thresh=thresh, hidden_dim=64,
                    hidden_dim_color=64, triplane_channels=16, triplane_resolution=64, triplane_wavelet_levels=2,
                    wavelet_type="bior6.8").to(dev)
    m.grid_size = H
    m.density_grid = torch.zeros(m.cascade, H ** 3, device=dev)
    m.density_bitfield = torch.zeros(m.cascade * H ** 3 // 8, dtype=torch.uint8, device=dev)
    m._morton_xyz = None
    m.reset_extra_state()
    return m


@pytest.mark.parametrize("tag", ["g32", "g128"])
def test_mark_untrained_grid_equals_reference(cuda, ref, tag):
    H, cascade = (int(v) for v in ref[f"{tag}/cfg"])
    m = _model(cuda, H, 10.0)
    m.mark_untrained_grid(ref["poses"], tuple(float(v) for v in ref["intrinsic"]))
    got = m.density_grid.cpu().numpy() == -1
    want = np.unpackbits(ref[f"{tag}/untrained"]).astype(bool).reshape(cascade, H ** 3)
    _, ambiguous = ogrid.mark_untrained_grid(ref["poses"], ref["intrinsic"], H, cascade, 1.5)
    assert np.array_equal(got[~ambiguous], want[~ambiguous])
    assert (got != want).sum() <= ambiguous.sum() < 1e-3 * want.size
    assert np.all(m.density_grid.cpu().numpy()[~got] == 0)


@pytest.mark.parametrize("tag", ["g32", "g128"])
def test_update_extra_state_equals_reference(cuda, ref, tag):
    H, cascade = (int(v) for v in ref[f"{tag}/cfg"])
    bound, thresh, dscale = (float(v) for v in ref[f"{tag}/cfg_f"])
    m = _model(cuda, H, thresh)
    untrained = np.unpackbits(ref[f"{tag}/untrained"]).astype(bool).reshape(cascade, H ** 3)
    m.density_grid.copy_(torch.from_numpy(np.where(untrained, np.float32(-1), np.float32(0))))
    order = torch.from_numpy(ogrid.reference_order(H))
    calls = []
    for name in STEPS:
        seed, zero_noise, local_step = (int(v) for v in ref[f"{tag}/{name}/seed"])
        m.step_counter.copy_(torch.from_numpy(ref[f"{tag}/ring"]))
        m.local_step = local_step
        if name == "part0":
            m.iter_density = 16
        blobs = blobs_of(ref, name)

        def density(x, blobs=blobs):
            calls.append(x.shape[0])
            return {"sigma": ogrid.blob_density(x, blobs), "geo_feat": None}
        m.density = density                                     # on the instance: the refresh must query THIS density
        d = ogrid.Draws(seed, bool(zero_noise))                 # replayed in the reference's call order
        prev = m.density_grid.cpu().numpy()
        if name.startswith("full"):
            draws = {"noise": [torch.from_numpy(d.rand((H ** 3, 3)))[order] for _ in range(cascade)]}
        else:
            N = H ** 3 // 4
            draws = {"noise": [], "coords": [], "occ_k": []}
            for cas in range(cascade):
                draws["coords"].append(torch.from_numpy(d.randint(0, H, (N, 3))).int())
                draws["occ_k"].append(torch.from_numpy(d.randint(0, int((prev[cas] > 0).sum()), (N,))))
                draws["noise"].append(torch.from_numpy(d.rand((2 * N, 3))))
        m.update_extra_state(draws=draws)
        grid = m.density_grid.cpu().numpy()
        rmean = float(ref[f"{tag}/{name}/mean_density"])
        if name != "part1":
            check_grid(ref, tag, name, grid)
        elif f"{tag}/{name}/grid" in ref.files:
            # jittered partial refresh: only cells drawn more than once may differ from the reference (its index_put_
            # keeps one candidate, this build's another); the oracle lists the candidates
            st = dict(density_grid=prev, step_counter=ref[f"{tag}/ring"], local_step=local_step, iter_density=17,
                      mean_count=0, mean_density=0.0)
            o = ogrid.update_extra_state(st, lambda x: ogrid.blob_density(x, blobs), ogrid.Draws(seed, False), H, cascade,
                                         bound, dscale, thresh)
            rgrid = ref[f"{tag}/{name}/grid"]
            for cas, (idx, sig) in enumerate(o["candidates"]):
                uniq, cnt = np.unique(idx, return_counts=True)
                multi = np.zeros(H ** 3, bool)
                multi[uniq[cnt > 1]] = True
                assert np.array_equal(grid[cas][~multi], rgrid[cas][~multi])
                srt = np.argsort(idx, kind="stable")
                sidx, ssig = idx[srt], sig[srt]
                for cell in np.nonzero(grid[cas] != rgrid[cas])[0]:
                    lo, hi = np.searchsorted(sidx, [cell, cell + 1])
                    assert grid[cas, cell] in np.maximum(prev[cas, cell] * np.float32(0.95), ssig[lo:hi])
            m.density_grid.copy_(torch.from_numpy(rgrid))
        assert abs(m.mean_density - rmean) <= 2e-6 * rmean or name == "part1", (m.mean_density, rmean)
        assert m.mean_count == int(ref[f"{tag}/{name}/mean_count"])
        assert m.iter_density == int(ref[f"{tag}/{name}/iter_density"]) and m.local_step == 0
        if name != "part1":
            bits = np.unpackbits(m.density_bitfield.cpu().numpy(), bitorder="little")
            rbits = np.unpackbits(ref[f"{tag}/{name}/bitfield"], bitorder="little")
            t = min(rmean, thresh)
            near = np.abs(grid.reshape(-1) - t) <= 1e-5 * t
            assert np.array_equal(bits[~near], rbits[~near])
            assert (bits != rbits).sum() <= near.sum() < 64
            # and the device-side threshold + packbits against the C oracle's packbits on this build's own mean
            assert np.array_equal(m.density_bitfield.cpu().numpy(), cref.packbits(grid, min(m.mean_density, thresh)))
    assert len(calls) == cascade * len(STEPS)                   # the overriding density() was the one queried


def test_grid_refresh_queries_an_overriding_density(cuda):
    """A subclass that overrides density() (custom activation / scale) must be what the refresh evaluates, not the stock
    network's fused sigma-only form (round-2 advisor finding)."""
    from trinerflet_amd.nerf.network import NeRFNetwork

    class Doubled(NeRFNetwork):
        def density(self, x):
            out = super().density(x)
            return {"sigma": out["sigma"] * 2, "geo_feat": out["geo_feat"]}

    kw = dict(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, density_thresh=10, hidden_dim=64, hidden_dim_color=64,
              triplane_channels=16, triplane_resolution=64, triplane_wavelet_levels=2, wavelet_type="bior6.8")
    torch.manual_seed(0)
    a = NeRFNetwork(**kw).to(cuda)
    b = Doubled(**kw).to(cuda)
    b.load_state_dict(a.state_dict())
    H = a.grid_size
    noise = [torch.rand(H ** 3, 3, device=cuda) for _ in range(a.cascade)]
    a.update_extra_state(draws={"noise": noise})
    b.update_extra_state(draws={"noise": noise})
    ga, gb = a.density_grid, b.density_grid
    assert float(ga.max()) > 0
    torch.testing.assert_close(gb, 2 * ga, rtol=1e-6, atol=0)


def test_partial_refresh_duplicate_picks_keep_the_larger_density(cuda):
    """A cell drawn more than once in one partial refresh (renderer.py:494-513: the reference's index assignment keeps an
    arbitrary candidate) keeps the LARGER candidate here -- deterministic, one of the reference's possible outcomes
    (INTEGRATION.md A.2); a NaN candidate propagates (amax) instead of being dropped by chance."""
    H = 32
    m = _model(cuda, H, 10.0)
    m.iter_density = 16                                            # partial refreshes from here on
    m.density_grid.fill_(0.0)
    m.density_grid[:, 100] = 1.0                                   # one occupied cell per cascade: every occupied pick hits it
    N = H ** 3 // 4
    coords = torch.zeros(N, 3, dtype=torch.int32)
    coords[:, 0] = 5                                               # every uniform pick is the cell (5, 0, 0)
    noise = torch.rand(2 * N, 3, generator=torch.Generator().manual_seed(0))
    draws = {"coords": [coords] * m.cascade, "occ_k": [torch.zeros(N, dtype=torch.int64)] * m.cascade,
             "noise": [noise] * m.cascade}
    seen = []

    def density(x):
        v = (x[:, 1] + 4.0).float()                                # positive, varies with the jitter: the candidates of a cell differ
        seen.append(v.clone())
        return {"sigma": v, "geo_feat": None}
    m.density = density
    m.update_extra_state(decay=1.0, draws=draws)
    from trinerflet_amd import raymarching
    cell = int(raymarching.morton3D(coords[:1].to(cuda))[0])
    for cas in range(m.cascade):
        cand = seen[cas] * m.density_scale
        assert float(m.density_grid[cas, cell]) == float(cand[:N].max())
        assert float(m.density_grid[cas, 100]) == max(1.0, float(cand[N:].max()))
    # every other cell untouched
    mask = torch.ones(H ** 3, dtype=torch.bool, device=cuda)
    mask[cell] = mask[100] = False
    assert float(m.density_grid[:, mask].abs().max()) == 0.0
