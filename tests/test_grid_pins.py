"""A12: the density-grid upkeep pinned by outputs of the reference's OWN methods.

tests/golden/grid_reference.npz was produced in the build container by tests/golden/make_golden_grid.py, which ran
reconstruction/nerf/renderer.py's mark_untrained_grid (:383-446) and update_extra_state (:448-542) -- imported
unmodified -- on an analytic density with seeded draws.  Here (CPU): oracle/grid.py, the numpy restatement, must
reproduce them -- grids bit for bit, bitfields bit for bit from the reference's own mean, mean_count exactly.
tests/test_grid_reference_gpu.py holds the HIP product to the same fixture."""
import os

import numpy as np
import pytest

from oracle import cref, grid as ogrid

STEPS = ["full0", "full1", "part0", "part1"]


@pytest.fixture(scope="module")
def ref(golden_dir):
    return np.load(os.path.join(golden_dir, "grid_reference.npz"))


def blobs_of(ref, name):
    return [tuple(float(v) for v in row) for row in ref["blobs_a" if name in ("full0", "part0") else "blobs_b"]]


def initial_state(ref, tag):
    H, cascade = (int(v) for v in ref[f"{tag}/cfg"])
    untrained = np.unpackbits(ref[f"{tag}/untrained"]).astype(bool).reshape(cascade, H ** 3)
    grid = np.where(untrained, np.float32(-1), np.float32(0))
    return dict(density_grid=grid, step_counter=ref[f"{tag}/ring"], local_step=0, iter_density=0, mean_count=0,
                mean_density=0.0)


def check_grid(ref, tag, name, grid):
    """Full arrays at H = 32; every 61st cell + float64 sums at H = 128."""
    if f"{tag}/{name}/grid" in ref.files:
        assert np.array_equal(grid, ref[f"{tag}/{name}/grid"]), (tag, name, int((grid != ref[f"{tag}/{name}/grid"]).sum()))
    else:
        assert np.array_equal(grid[:, ::61], ref[f"{tag}/{name}/grid_every61"]), (tag, name)
        np.testing.assert_allclose(grid.astype(np.float64).sum(1), ref[f"{tag}/{name}/grid_sum"], rtol=1e-12)
        np.testing.assert_allclose(np.abs(grid.astype(np.float64)).sum(1), ref[f"{tag}/{name}/grid_abs_sum"], rtol=1e-12)


@pytest.mark.parametrize("tag", ["g32", "g128"])
def test_mark_untrained_grid_oracle_equals_reference(ref, tag):
    H, cascade = (int(v) for v in ref[f"{tag}/cfg"])
    bound = float(ref[f"{tag}/cfg_f"][0])
    want = np.unpackbits(ref[f"{tag}/untrained"]).astype(bool).reshape(cascade, H ** 3)
    got, ambiguous = ogrid.mark_untrained_grid(ref["poses"], ref["intrinsic"], H, cascade, bound)
    assert ambiguous.mean() < 1e-3
    assert np.array_equal(got[~ambiguous], want[~ambiguous])
    assert (got != want).sum() <= ambiguous.sum()
    assert 0.02 < want.mean() < 0.5          # the fixture's cameras leave part of the volume unseen, not all of it


@pytest.mark.parametrize("tag", ["g32", "g128"])
def test_update_extra_state_oracle_equals_reference(ref, tag):
    H, cascade = (int(v) for v in ref[f"{tag}/cfg"])
    bound, thresh, dscale = (float(v) for v in ref[f"{tag}/cfg_f"])
    st = initial_state(ref, tag)
    for name in STEPS:
        seed, zero_noise, local_step = (int(v) for v in ref[f"{tag}/{name}/seed"])
        st["local_step"] = local_step
        st["step_counter"] = ref[f"{tag}/ring"]
        if name == "part0":
            st["iter_density"] = 16
        blobs = blobs_of(ref, name)
        new = ogrid.update_extra_state(st, lambda x: ogrid.blob_density(x, blobs), ogrid.Draws(seed, bool(zero_noise)), H,
                                       cascade, bound, dscale, thresh)
        rmean = float(ref[f"{tag}/{name}/mean_density"])
        if name != "part1":
            check_grid(ref, tag, name, new["density_grid"])
            grid_for_bits = new["density_grid"]
        else:
            # jittered partial refresh: a cell drawn more than once keeps ONE of its candidates (which one is the
            # reference's index_put_ race, renderer.py:515); every reference value must be the EMA of a candidate
            rgrid = ref[f"{tag}/{name}/grid"] if f"{tag}/{name}/grid" in ref.files else None
            if rgrid is not None:
                prev = st["density_grid"]
                for cas, (idx, sig) in enumerate(new["candidates"]):
                    differs = np.nonzero(new["density_grid"][cas] != rgrid[cas])[0]
                    assert differs.size < 0.2 * idx.size
                    order = np.argsort(idx, kind="stable")
                    sidx, ssig = idx[order], sig[order]
                    for cell in differs:
                        lo, hi = np.searchsorted(sidx, [cell, cell + 1])
                        assert hi - lo >= 2, (cas, cell)                      # only multiply drawn cells may differ
                        cands = np.maximum(prev[cas, cell] * np.float32(0.95), ssig[lo:hi])
                        assert rgrid[cas, cell] in cands
                grid_for_bits = rgrid
                new["density_grid"] = rgrid.copy()                            # continue from the reference's state
            else:
                sub, rsub = new["density_grid"][:, ::61], ref[f"{tag}/{name}/grid_every61"]
                assert (sub != rsub).mean() < 0.2
                grid_for_bits = None
        assert abs(new["mean_density"] - rmean) <= 2e-6 * rmean or name == "part1"
        assert new["mean_count"] == int(ref[f"{tag}/{name}/mean_count"])
        assert new["iter_density"] == int(ref[f"{tag}/{name}/iter_density"]) and new["local_step"] == 0
        if grid_for_bits is not None:
            # threshold rule + packbits on the reference's own (fp32) mean: bit for bit
            assert np.array_equal(cref.packbits(grid_for_bits, min(rmean, thresh)), ref[f"{tag}/{name}/bitfield"])
        st = {k: new[k] for k in ("density_grid", "step_counter", "local_step", "iter_density", "mean_count", "mean_density")}


def test_fixture_exercises_both_threshold_regimes_and_the_ema(ref):
    assert float(ref["g32/full0/mean_density"]) < float(ref["g32/cfg_f"][1])        # threshold = mean_density
    assert float(ref["g128/full0/mean_density"]) > float(ref["g128/cfg_f"][1])      # threshold = density_thresh
    g0, g1 = ref["g32/full0/grid"], ref["g32/full1/grid"]
    decayed = (g1 == g0 * np.float32(0.95)) & (g0 > 0)
    fresh = (g1 != g0 * np.float32(0.95)) & (g1 > 0)
    assert decayed.sum() > 500 and fresh.sum() > 500                                # both arms of max(grid*decay, new)
    assert int(ref["g32/full0/mean_count"]) == int(ref["g32/ring"][:5, 0].sum() / 5)
    assert int(ref["g32/full1/mean_count"]) == int(ref["g32/ring"][:16, 0].sum() / 16)
