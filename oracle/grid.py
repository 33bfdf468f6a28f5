"""TEST INFRASTRUCTURE (the checker, never the product): CPU restatement in numpy of the density-grid upkeep of
the reference's renderer -- SURVEY.md 8(a) row A12 host logic.

    mark_untrained_grid   reconstruction/nerf/renderer.py:383-446
    update_extra_state    reconstruction/nerf/renderer.py:448-542   (full refresh :459-489, partial refresh :490-515,
                                                                      EMA :524-527, threshold + packbits :531-534,
                                                                      mean_count from the step_counter ring :536-541)

The Morton / packbits kernels underneath are the C oracle's (oracle/trinerflet_oracle.c, raymarching.cu:214-300).
Pinned by tests/golden/grid_reference.npz, which tests/golden/make_golden_grid.py produced by RUNNING the reference's
own two methods (imported from /root/reference, unmodified) with the random draws below and the analytic density
below substituted for torch's RNG and the network (tests/test_grid_pins.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

from . import cref


class Draws:
    """The random numbers of one update_extra_state call, in the order the reference draws them (per cascade: full
    refresh torch.rand_like [H^3,3] :484; partial refresh torch.randint(0,H,(N,3)) :494, torch.randint(0,n_occ,[N]) :499,
    torch.rand_like [2N,3] :509), from a numpy generator so that the generator script, this restatement and the GPU
    test regenerate the same values from a seed stored in the fixture.  zero_noise: every jitter draw is 0.5, i.e. the
    cell centre itself (a cell drawn twice in a partial refresh then carries one value whichever write wins)."""

    def __init__(self, seed, zero_noise=False):
        self.rng = np.random.default_rng(int(seed))
        self.zero_noise = zero_noise

    def rand(self, shape):
        x = self.rng.random(tuple(shape), dtype=np.float32)      # drawn even when unused: same stream either way
        return np.full(tuple(shape), 0.5, np.float32) if self.zero_noise else x

    def randint(self, lo, hi, shape):
        return self.rng.integers(int(lo), int(hi), tuple(shape), dtype=np.int64)


def blob_density(x, blobs):
    """Analytic density used by the A12 fixtures: sum over blobs (cx, cy, cz, peak, falloff) of
    max(peak - falloff * |x - c|^2, 0), written with +, -, * and a clamp only -- every operation correctly rounded in
    fp32 on the CPU (numpy and torch) and on the GPU (one torch kernel per operation, so nothing is contracted into an
    FMA): the three parties evaluate bit-identical densities.  x: [n,3] float32 numpy array or torch tensor."""
    x0, x1, x2 = x[:, 0], x[:, 1], x[:, 2]
    total = None
    for cx, cy, cz, peak, fall in blobs:
        d0, d1, d2 = x0 - cx, x1 - cy, x2 - cz
        q = d0 * d0 + d1 * d1
        q = q + d2 * d2
        v = (peak - q * fall).clip(min=0)
        total = v if total is None else total + v
    return total


def mark_untrained_grid(poses, intrinsic, H, cascade, bound, S=64):
    """renderer.py:383-446.  Returns (untrained [cascade, H^3] bool in Morton order -- the cells the reference sets to
    -1 -- and ambiguous [cascade, H^3] bool: cells for which some camera's visibility test is decided by less than
    1e-5 (evaluated in float64), where a matmul with another summation order may legitimately decide otherwise)."""
    poses = np.asarray(poses, np.float32)
    fx, fy, cx, cy = (float(v) for v in intrinsic)
    ax = np.arange(H, dtype=np.int32)
    coords = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)       # custom_meshgrid = 'ij' (:413)
    indices = cref.morton3D(coords).astype(np.int64)
    world = (np.float32(2) * coords.astype(np.float32) / np.float32(H - 1) - np.float32(1))          # :415
    count = np.zeros((cascade, H ** 3), np.int64)
    ambiguous = np.zeros((cascade, H ** 3), bool)
    for cas in range(cascade):
        b = min(2 ** cas, bound)
        hgs = b / H
        cw = world * np.float32(b - hgs)                                                                # :422
        for head in range(0, poses.shape[0], S):
            P = poses[head:head + S]
            cam = np.einsum("snk,skj->snj", cw[None] - P[:, None, :3, 3], P[:, :3, :3])                # :429-430
            z = cam[:, :, 2]
            mx = np.float32(cx / fx) * z + np.float32(hgs * 2) - np.abs(cam[:, :, 0])                   # :434-435, > 0 = inside
            my = np.float32(cy / fy) * z + np.float32(hgs * 2) - np.abs(cam[:, :, 1])
            mask = (z > 0) & (mx > 0) & (my > 0)
            count[cas, indices] += mask.sum(0)
            camd = np.einsum("snk,skj->snj", cw[None].astype(np.float64) - P[:, None, :3, 3].astype(np.float64),
                             P[:, :3, :3].astype(np.float64))
            zd = camd[:, :, 2]
            margins = np.stack([zd, cx / fx * zd + hgs * 2 - np.abs(camd[:, :, 0]),
                                cy / fy * zd + hgs * 2 - np.abs(camd[:, :, 1])])
            # a camera's verdict is fragile when its smallest margin is within 1e-5 of zero
            fragile = np.abs(margins.min(0)) < 1e-5
            ambiguous[cas, indices] |= fragile.any(0)
    return count == 0, ambiguous


def update_extra_state(state, density, draws, H, cascade, bound, density_scale=1.0, density_thresh=10.0, decay=0.95):
    """renderer.py:448-542 on `state` = dict(density_grid [cascade,H^3] f32, step_counter [16,2] i32, local_step,
    iter_density, mean_count, mean_density); returns the new state (+ density_bitfield, + `candidates`: per cascade the
    list (indices, sigmas) that was assigned into tmp_grid, for the duplicate-pick check of the partial refresh).
    density: callable [n,3] f32 -> [n] f32 (self.density(x)['sigma']); draws: a Draws."""
    grid = np.array(state["density_grid"], np.float32, copy=True)
    tmp_grid = -np.ones_like(grid)                                                                      # :456
    candidates = []
    if state["iter_density"] < 16:                                                                      # full refresh :459
        ax = np.arange(H, dtype=np.int32)
        coords = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)                    # S = 128: one block
        indices = cref.morton3D(coords).astype(np.int64)                                                # :472
        xyzs = np.float32(2) * coords.astype(np.float32) / np.float32(H - 1) - np.float32(1)            # :473
        for cas in range(cascade):
            b = min(2 ** cas, bound)
            hgs = b / H
            cas_xyzs = xyzs * np.float32(b - hgs)                                                       # :480
            cas_xyzs = cas_xyzs + (draws.rand(cas_xyzs.shape) * np.float32(2) - np.float32(1)) * np.float32(hgs)   # :482
            sig = density(cas_xyzs).reshape(-1).astype(np.float32) * np.float32(density_scale)          # :484-485
            tmp_grid[cas, indices] = sig                                                                # :487
            candidates.append((indices, sig))
    else:                                                                                               # partial refresh :491
        N = H ** 3 // 4
        for cas in range(cascade):
            coords = draws.randint(0, H, (N, 3)).astype(np.int32)                                       # :494
            indices = cref.morton3D(coords).astype(np.int64)                                            # :495
            occ = np.nonzero(grid[cas] > 0)[0]                                                          # :497
            pick = draws.randint(0, occ.shape[0], (N,))                                                 # :498
            occ_indices = occ[pick]                                                                     # :499
            occ_coords = cref.morton3D_invert(occ_indices.astype(np.int32))                             # :500
            indices = np.concatenate([indices, occ_indices])                                            # :502
            coords = np.concatenate([coords, occ_coords])
            xyzs = np.float32(2) * coords.astype(np.float32) / np.float32(H - 1) - np.float32(1)        # :505
            b = min(2 ** cas, bound)
            hgs = b / H
            cas_xyzs = xyzs * np.float32(b - hgs)
            cas_xyzs = cas_xyzs + (draws.rand(cas_xyzs.shape) * np.float32(2) - np.float32(1)) * np.float32(hgs)
            sig = density(cas_xyzs).reshape(-1).astype(np.float32) * np.float32(density_scale)
            tmp_grid[cas, indices] = sig                                                                # :515 (a cell drawn twice: one of its values)
            candidates.append((indices, sig))
    valid = (grid >= 0) & (tmp_grid >= 0)                                                               # :524
    grid[valid] = np.maximum(grid[valid] * np.float32(decay), tmp_grid[valid])                          # :525
    mean_density = float(np.mean(np.clip(grid, 0, None), dtype=np.float64))                             # :526 (.item())
    thresh = min(mean_density, density_thresh)                                                          # :531
    bitfield = cref.packbits(grid, thresh)                                                              # :532
    total_step = min(16, state["local_step"])                                                           # :535
    mean_count = state["mean_count"]
    if total_step > 0:
        mean_count = int(int(np.asarray(state["step_counter"])[:total_step, 0].sum()) / total_step)     # :537
    return dict(density_grid=grid, density_bitfield=bitfield, mean_density=mean_density,
                iter_density=state["iter_density"] + 1, mean_count=mean_count, local_step=0,
                step_counter=np.asarray(state["step_counter"]), candidates=candidates, thresh=thresh)


def reference_order(H):
    """For a full refresh: position j of every Morton-ordered cell i in the reference's evaluation order (the 'ij'
    meshgrid, j = (x*H + y)*H + z) -- noise_in_morton_order = noise_in_reference_order[reference_order(H)]."""
    c = cref.morton3D_invert(np.arange(H ** 3, dtype=np.int32)).astype(np.int64)
    return (c[:, 0] * H + c[:, 1]) * H + c[:, 2]
