"""The C oracle against its own frozen outputs (tests/golden/oracle_kernels.npz, generator make_golden_oracle.py):
any edit of oracle/trinerflet_oracle.c that changes a kernel's result shows up here, on the CPU, before the GPU
tests (tests/test_native_backends_gpu.py) compare the HIP kernels with the same file."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import cref

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_golden_oracle", os.path.join(HERE, "golden", "make_golden_oracle.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "oracle_kernels.npz"))


def test_sh4_equals_the_reference_formulas(golden_dir):
    """oracle SH-4 == the reference kernel's own statements evaluated in fp32 (tests/golden/sh_reference.npz, made by
    make_golden_sh.py from shencoder.cu where it lies): bit for bit -- A4 is pinned by execution of the reference's
    arithmetic, not only by restatement."""
    g = np.load(os.path.join(golden_dir, "sh_reference.npz"))
    assert np.array_equal(cref.sh4(g["dirs"]), g["out"][:, :16])
    # and the 64 functions the fixture holds are orthonormal on the sphere (sanity of the fixture itself)
    rng = np.random.default_rng(0)
    d = rng.standard_normal((200000, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    assert np.allclose(cref.sh4(d.astype(np.float32)).astype(np.float64).T @ cref.sh4(d.astype(np.float32)) * 4 * np.pi / len(d),
                       np.eye(16), atol=0.03)


def test_sh_and_grid(fx):
    np.testing.assert_array_equal(cref.sh4(fx["sh/dirs"]), fx["sh/out"])
    ax = np.arange(128, dtype=np.int32)
    coords = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    assert np.array_equal(gen.sha(cref.morton3D(coords)), fx["grid/morton_sha"])
    assert np.array_equal(cref.morton3D(fx["grid/sample_coords"]), fx["grid/sample_codes"])
    grid = np.random.default_rng(int(fx["grid/seed"])).standard_normal((2, 128 ** 3)).astype(np.float32)
    assert np.array_equal(gen.sha(cref.packbits(grid, float(fx["grid/thresh"]))), fx["grid/packbits_sha"])


@pytest.mark.parametrize("cfg", ["plain", "perturb", "budget"])
def test_march(fx, cfg):
    o, d = fx["rays/o"], fx["rays/d"]
    aabb = np.array([-gen.BOUND] * 3 + [gen.BOUND] * 3, np.float32)
    nears, fars = cref.near_far_from_aabb(o, d, aabb, 0.2)
    assert np.array_equal(nears, fx["rays/nears"]) and np.array_equal(fars, fx["rays/fars"])
    M = int(fx[f"march/{cfg}/M"])
    x, dd, dl, rr, cnt = cref.march_rays_train(o, d, gen.BOUND, fx["bitfield"], gen.CAS, gen.HG, nears, fars,
                                               fx[f"march/{cfg}/noises"], M, 0.0, gen.MAX_STEPS)
    assert np.array_equal(rr, fx[f"march/{cfg}/rays"]) and np.array_equal(cnt, fx[f"march/{cfg}/counter"])
    m = min(int(cnt[0]), M)
    assert np.array_equal(gen.sha(x[:m]), fx[f"march/{cfg}/sha_xyzs"])
    assert np.array_equal(gen.sha(dl[:m]), fx[f"march/{cfg}/sha_deltas"])
    rows = fx[f"march/{cfg}/sub_rows"]
    assert np.array_equal(x[rows], fx[f"march/{cfg}/sub_xyzs"]) and np.array_equal(dl[rows], fx[f"march/{cfg}/sub_deltas"])
    if cfg == "budget":
        dropped = (rr[:, 2] > 0) & (rr[:, 1] + rr[:, 2] > M)
        assert dropped.any() and int(cnt[0]) > M
