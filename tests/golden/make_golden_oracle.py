"""Generates tests/golden/oracle_kernels.npz: outputs of the CPU oracle (oracle/trinerflet_oracle.c) for the CUDA-only
kernels of the reference (raymarching.cu, shencoder.cu), frozen so that the GPU box compares the HIP kernels against
committed data and the CPU suite notices any drift of the oracle itself (SURVEY.md 8(c) "minimum fixture set"):

    python tests/golden/make_golden_oracle.py

THESE VECTORS ARE NOT REFERENCE OUTPUTS.  The reference's kernels cannot run here (CUDA-only, no CPU path, SURVEY
F4/F6); they are the line-by-line C restatement's outputs ("parity unpinned by execution").

  sh/{dirs, out}                           F-SH     64 unit directions -> 16 values (shencoder.cu:50-68)
  rays/{o, d, nears, fars}, bitfield       512 rays of 4 cameras (+ axis-parallel, + a miss), sphere-shell occupancy
  march/<cfg>/{noises, rays, counter, sha_*, sub_*}   F-MARCH  cfg in {plain, perturb, budget}: per-ray (id, offset,
                                           count) exact, SHA-256 of the xyzs / dirs / deltas bytes, full rows of every
                                           16th ray
  comp/{sigmas_seed.., ws, depth, image, g_*, sub_gs, sub_gc, sum_gs, sum_gc}   F-COMP forward + backward
  infer/{image, depth, ws, alive_sha, n_alive}       F-INFER  the alive-ray loop on 1024 rays with an analytic field
  grid/{morton_sha, invert_ok, packbits_sha, ...}    F-GRID   all 128^3 Morton codes, packbits of a seeded grid
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import cref  # noqa: E402

BOUND, CAS, HG, MAX_STEPS = 1.5, 2, 128, 1024


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def cameras(n):
    poses = []
    for a in range(n):
        th, ph = 0.5 + 0.3 * a, 1.7 * a
        eye = 4.0 * np.array([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)])
        fwd = -eye / np.linalg.norm(eye)
        right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        pose = np.eye(4, dtype=np.float32)
        pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, up, fwd, eye
        poses.append(pose)
    return np.stack(poses).astype(np.float32)


def shell_bitfield(r_out, r_in):
    ax = np.arange(HG, dtype=np.int32)
    coords = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    idx = cref.morton3D(coords)
    grid = np.zeros((CAS, HG ** 3), np.float32)
    for c in range(CAS):
        s = min(2.0 ** c, BOUND)
        rad = np.linalg.norm(((coords + 0.5) / HG * 2 - 1) * s, axis=1)
        grid[c, idx] = ((rad < r_out) & (rad >= r_in)).astype(np.float32)
    return cref.packbits(grid, 0.5)


def analytic_field(x, d):
    """Deterministic float32 numpy field, evaluated identically by the generator and by the tests."""
    s = (8.0 * np.exp(-4.0 * (x.astype(np.float32) ** 2).sum(-1))).astype(np.float32)
    c = (0.5 + 0.5 * np.sin(3.0 * x + d)).astype(np.float32)
    return s, c


def main():
    rng = np.random.default_rng(2024)
    out = {}
    # ---- F-SH
    dirs = rng.standard_normal((64, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    dirs[:3] = np.eye(3, dtype=np.float32)
    out["sh/dirs"], out["sh/out"] = dirs, cref.sh4(dirs)
    # ---- rays
    N = 512
    poses = cameras(4)
    pix = rng.integers(0, 4 * 64 * 64, N)
    o, d = cref.get_rays(poses, np.array([90.0, 90.0, 32.0, 32.0], np.float32), 64, 64, pix)
    d[:4] = np.array([[0, 1, 0], [1, 0, 0], [0, 0, -1], [0.6, 0.8, 0.0]], np.float32)
    o[:4] = np.array([[0.1, -3, 0.2], [-3, 0.3, 0.1], [0.2, 0.1, 3], [-2.4, -3.2, 0.05]], np.float32)
    o[4:6] += 10.0
    aabb = np.array([-BOUND] * 3 + [BOUND] * 3, np.float32)
    nears, fars = cref.near_far_from_aabb(o, d, aabb, 0.2)
    bf = shell_bitfield(0.8, 0.55)
    out.update({"rays/o": o, "rays/d": d, "rays/nears": nears, "rays/fars": fars, "bitfield": bf})
    # ---- F-MARCH
    sub = np.arange(0, N, 16)
    full = None
    for cfg in ("plain", "perturb", "budget"):
        noises = np.zeros(N, np.float32) if cfg == "plain" else rng.random(N).astype(np.float32)
        M = N * MAX_STEPS
        if cfg == "budget":
            M = full // 2 + (128 - (full // 2) % 128)            # drops the last rays (raymarching.cu:405-422)
        x, dd, dl, rr, cnt = cref.march_rays_train(o, d, BOUND, bf, CAS, HG, nears, fars, noises, M, 0.0, MAX_STEPS)
        total = int(cnt[0])
        if cfg == "plain":
            full = total
        m = min(total, M)
        out.update({f"march/{cfg}/noises": noises, f"march/{cfg}/rays": rr, f"march/{cfg}/counter": cnt,
                    f"march/{cfg}/M": np.array(M), f"march/{cfg}/sha_xyzs": sha(x[:m]), f"march/{cfg}/sha_dirs": sha(dd[:m]),
                    f"march/{cfg}/sha_deltas": sha(dl[:m])})
        rows = np.concatenate([np.arange(rr[i, 1], rr[i, 1] + rr[i, 2]) for i in sub
                               if rr[i, 2] > 0 and rr[i, 1] + rr[i, 2] <= M] or [np.zeros(0, np.int64)]).astype(np.int64)
        out.update({f"march/{cfg}/sub_rows": rows, f"march/{cfg}/sub_xyzs": x[rows], f"march/{cfg}/sub_deltas": dl[rows]})
        if cfg == "perturb":
            keep = (x, dd, dl, rr, total)
    # ---- F-COMP on the perturbed march
    x, dd, dl, rr, total = keep
    Mc = total + (128 - total % 128)
    g = np.random.default_rng(7)
    sig = np.exp(g.standard_normal(Mc) * 2.0).astype(np.float32)
    rgb = g.random((Mc, 3)).astype(np.float32)
    rr2 = rr.copy()
    rr2[9, 1] = Mc                                               # an overflowing ray: zero outputs, no gradient
    ws, dep, img = cref.composite_rays_train_forward(sig, rgb, dl[:Mc], rr2, 1e-4)
    gws = g.standard_normal(N).astype(np.float32)
    gimg = g.standard_normal((N, 3)).astype(np.float32)
    gs, gc = cref.composite_rays_train_backward(gws, gimg, sig, rgb, dl[:Mc], rr2, ws, img, 1e-4)
    rows = out["march/perturb/sub_rows"]
    out.update({"comp/M": np.array(Mc), "comp/seed": np.array(7), "comp/rays": rr2, "comp/ws": ws, "comp/depth": dep,
                "comp/image": img, "comp/sub_gs": gs[rows], "comp/sub_gc": gc[rows],
                "comp/sum_gs": np.array(gs.astype(np.float64).sum()), "comp/sum_abs_gs": np.array(np.abs(gs).astype(np.float64).sum()),
                "comp/sum_gc": gc.astype(np.float64).sum(0)})
    # ---- F-INFER: the alive loop (renderer.py:324-374) on 1024 rays, analytic field
    Ni = 1024
    pix = rng.integers(0, 4 * 64 * 64, Ni)
    oi, di = cref.get_rays(poses, np.array([90.0, 90.0, 32.0, 32.0], np.float32), 64, 64, pix)
    ni, fi = cref.near_far_from_aabb(oi, di, aabb, 0.2)
    ws, dep, img = np.zeros(Ni, np.float32), np.zeros(Ni, np.float32), np.zeros((Ni, 3), np.float32)
    alive = np.arange(Ni, dtype=np.int32)
    rt = ni.copy()
    step, n_hist, h = 0, [], hashlib.sha256()
    while step < MAX_STEPS:
        n_alive = alive.shape[0]
        if n_alive <= 0:
            break
        n_step = max(min(Ni // n_alive, 8), 1)
        xs, ds, ls = cref.march_rays(n_alive, n_step, alive, rt, oi, di, BOUND, bf, CAS, HG, ni, fi,
                                     np.zeros(n_alive, np.float32), 128, 0.0, MAX_STEPS)
        s, c = analytic_field(xs, ds)
        cref.composite_rays(n_alive, n_step, alive, rt, s, c, ls, ws, dep, img, 1e-2)
        alive = alive[alive >= 0]
        n_hist.append(alive.shape[0])
        h.update(alive.tobytes())
        step += n_step
    out.update({"infer/o": oi, "infer/d": di, "infer/nears": ni, "infer/fars": fi, "infer/ws": ws, "infer/depth": dep,
                "infer/image": img, "infer/n_alive": np.array(n_hist, np.int64),
                "infer/alive_sha": np.frombuffer(h.digest(), np.uint8).copy()})
    # ---- F-GRID
    ax = np.arange(HG, dtype=np.int32)
    coords = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    codes = cref.morton3D(coords)
    assert np.array_equal(np.sort(codes), np.arange(HG ** 3)) and np.array_equal(cref.morton3D_invert(codes), coords)
    grid = np.random.default_rng(5).standard_normal((CAS, HG ** 3)).astype(np.float32)
    out.update({"grid/morton_sha": sha(codes), "grid/seed": np.array(5), "grid/thresh": np.array(0.1, np.float32),
                "grid/packbits_sha": sha(cref.packbits(grid, 0.1)),
                "grid/sample_coords": coords[::40009], "grid/sample_codes": codes[::40009]})
    path = os.path.join(HERE, "oracle_kernels.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
    for k in ("march/plain/counter", "march/perturb/counter", "march/budget/counter", "march/budget/M", "infer/n_alive"):
        print(k, out[k])


if __name__ == "__main__":
    main()
