#!/usr/bin/env python
"""bench.py -- train rays/sec of the TriNeRFLet hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full optimisation step of the base configuration (BASELINE.json configs[2]/[3]:
3 planes x 32 channels x 2048^2, wavelet scale 32 = 5 levels of bior6.8, hidden 64/64, 60 000 rays, fp16
planes + fp16 MFMA MLP, fp32 master parameters): rebuild planes (IDWT) -> density-grid refresh every 16
steps -> near/far -> march -> fused field -> composite -> loss -> backward of all of it -> fused Adam+L1.
Synthetic inputs per SURVEY.md 8(d): 100 hemisphere cameras (800x800), seeded ray draw, analytic solid-sphere
occupancy r=0.8 re-imposed after each grid refresh (the refresh itself runs and is timed), seeded field.
Weak scaling: every rank processes its own 60 000 rays per step; value = rays of all ranks / max-rank time.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (the dominant kernel: the
fused Adam+L1 pass over the 402 M wavelet coefficients, HBM-bound, 28 B/parameter) and `cpu_baseline`.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)

WORKLOADS = {
    # name: (channels, resolution, wavelet scale, hidden, rays, lambda)
    "base": (32, 2048, 32, 64, 60000, 0.4),
    "small": (16, 1024, 16, 64, 60000, 0.2),
    "large": (48, 2048, 32, 128, 60000, 0.6),
    "tiny": (16, 256, 4, 64, 4096, 0.2),
}


def build(workload, device, dist_mode, plane_dtype=None, **ts_kwargs):
    from trinerflet_amd import synthetic
    from trinerflet_amd.nerf.network import NeRFNetwork
    from trinerflet_amd.train import TrainStep
    C, R, scale, H, N, lam = WORKLOADS[workload]
    extra = {} if plane_dtype is None else {"plane_dtype": plane_dtype}
    model = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, density_scale=1, min_near=0.2,
                        density_thresh=10, bg_radius=-1, hidden_dim=H, hidden_dim_color=H,
                        triplane_channels=C, triplane_resolution=R, triplane_wavelet_levels=scale,
                        wavelet_type="bior6.8", **extra).to(device)
    synthetic.init_field_parameters(model, seed=0)
    ts = TrainStep(model, lr=1e-2, wavelet_regularization=lam, iters=40000, warmup_steps=0, fp16=True,
                   background_color=0.0, dist_mode=dist_mode, **ts_kwargs)
    ts.prefetch_at = os.environ.get("TNL_PREFETCH_AT", ts.prefetch_at)   # "auto" | "bwd" | "adam" (experiments)
    if os.environ.get("TNL_SIDE_CUS"):        # experiments: CUs the side stream may use
        ts.side_cus = int(os.environ["TNL_SIDE_CUS"])
    if os.environ.get("TNL_NO_OVERLAP"):      # experiments: march + tile sort in order on the launch stream (kernels alone)
        ts.overlap_march = False
    bitfield = torch.from_numpy(synthetic.sphere_bitfield(128, model.cascade, 1.5, 0.8, 0.0)).to(device)
    model.density_bitfield.copy_(bitfield)
    return model, ts, bitfield, N


def make_batches(n_batches, N, rank, device):
    from trinerflet_amd import synthetic
    out = []
    poses = synthetic.hemisphere_poses(100, seed=0)
    for b in range(n_batches):
        rng = np.random.default_rng(1000 * rank + b)
        flat = rng.permutation(np.unique(rng.integers(0, 100 * 800 * 800, size=N + N // 8 + 16)))[:N]  # distinct pixels
        pix = np.stack([flat // (800 * 800), flat % (800 * 800)], -1)
        o, d = synthetic.get_rays(poses, pix)
        noise = rng.random(N).astype(np.float32)
        gt = synthetic.target_colors(d)
        out.append(tuple(torch.from_numpy(a).to(device) for a in (o, d, gt, noise)))
    return out


def one_step(model, ts, bitfield, batch, mean_count, next_batch=None):
    o, d, gt, noise = batch

    def reimpose():  # analytic occupancy re-imposed after the (timed) refresh; fixed sample budget
        model.density_bitfield.copy_(bitfield)
        model.mean_count = mean_count
    ts.post_refresh = reimpose
    # the following batch (a loader has it ready) lets its march start underneath this step's kernels
    nxt = None if next_batch is None else (next_batch[0], next_batch[1], next_batch[3])
    return ts.step(o, d, gt, noises=noise, next_rays=nxt)


def cpu_baseline(workload):
    """Reference operator set on the host cores, bounded sample (see oracle/torch_baseline.py)."""
    from oracle import torch_baseline as tb
    C, R, scale, H, N, lam = WORKLOADS[workload]
    # torch's CPU kernels stop scaling (and then regress) well before the 256 hardware threads of the GPU box's
    # host: 32 threads measured fastest there (8: 1.38 s, 16: 1.23 s, 32: 1.10 s, 64: 1.92 s for the same sample)
    cores = min(os.cpu_count() or 1, 32)
    # dense part at 1/4 of the plane area (R/2), per-ray part on N/10 rays; both scale linearly
    Rs, Ns = max(R // 2, 64 * 2), max(N // 10, 64)
    ss = max(scale // 2, 2)
    t = tb.time_step(C, Rs, ss, H, Ns, lam=lam, threads=cores)
    dense = t["dense_s"] * (R / Rs) ** 2
    ray = t["ray_s"] * (N / Ns)
    return {"value": N / (dense + ray), "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"torch-CPU fp32 step: dense part (IDWT fwd+bwd, L1, Adam) at R={Rs} scaled x{(R / Rs) ** 2:.0f}; "
                      f"per-ray part (512 uniform steps/ray, renderer.run semantics) on {Ns} rays scaled x{N / Ns:.0f}; "
                      f"measured dense {t['dense_s']:.2f}s ray {t['ray_s']:.2f}s"}


PMC_STATIC = os.path.join(ROOT, "profiles", "r02_pmc_adam.json")


def pmc_traffic_static(workload, world):
    """HBM bytes of the step's k_adam_l1 launches from the committed rocprofv3 PMC passes (tools/pmc_summary.py)."""
    if workload != "base" or world != 1 or not os.path.exists(PMC_STATIC):
        return None, None
    return json.load(open(PMC_STATIC))["hbm_bytes_per_launch"], os.path.relpath(PMC_STATIC, ROOT) + " (static)"


def pmc_traffic_live(workload, launches_per_step):
    """HBM bytes per k_adam_l1 launch measured NOW: two child runs of this script under `rocprofv3 --pmc` (FETCH_SIZE and
    WRITE_SIZE in separate passes, no trace domain, the program itself after `--`), 2 timed steps each, summed over the
    step's launches with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts half of a wide streaming
    read): bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Returns (bytes per step, note) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    tot = {}
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(td, counter)
            cmd = [prof, "--pmc", counter, "-d", out, "--output-format", "csv", "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "2", "--warmup", "0", "--pmc-child"]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=td)
            except Exception as e:                                   # noqa: BLE001
                return None, f"rocprofv3 child failed: {e}"
            if r.returncode != 0:
                return None, f"rocprofv3 child rc={r.returncode}: {r.stderr[-300:]}"
            vals = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] == counter and ("k_adam_l1<true" in row["Kernel_Name"] or
                                                               "k_adam_l1_live" in row["Kernel_Name"]):
                            vals.append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
            vals = [v for _, v in sorted(vals)]
            n = len(vals) // launches_per_step
            if n < 2:
                return None, f"no k_adam_l1<true> / k_adam_l1_live dispatches in the {counter} pass"
            last = vals[(n - 2) * launches_per_step:n * launches_per_step]     # the child's last two steps (ROI steps)
            tot[counter] = sum(last) / 2.0
    return (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0, \
        "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run, (2*FETCH+WRITE)*1024 per step"


def time_steps(model, ts, bitfield, batches, mean_count, steps, setup=4):
    """Seconds per step of `steps` steps after `setup` untimed ones (secondary figures of the bench line)."""
    nb = len(batches)
    for i in range(setup):
        one_step(model, ts, bitfield, batches[i % nb], mean_count, batches[(i + 1) % nb])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        one_step(model, ts, bitfield, batches[(setup + i) % nb], mean_count, batches[(setup + i + 1) % nb])
    ts.flush_deferred()      # deferred optimiser work of these steps belongs to them
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def variant_ms(workload, device, bitfield_np_unused, batches, mean_count, steps, **kw):
    """ms per step of a variant configuration (whole planes instead of the occupancy window, fp32 planes): a second
    model + TrainStep, a whole density-grid period of set-up, then `steps` timed steps that contain no refresh."""
    model, ts, bitfield, _ = build(workload, device, None, **kw)
    model.mean_count = mean_count
    t = time_steps(model, ts, bitfield, batches, mean_count, steps, setup=17)   # past the refresh at step 16
    del model, ts
    gc.collect()
    torch.cuda.empty_cache()
    return t * 1e3


def inference_figure(model, device, max_steps=4096, images=3):
    """BASELINE config 5's `--test` render: 800 x 800 rays of one pose through run_cuda's inference branch at
    max_steps = 4096 (renderer.py:324-374), the trained-state planes of the benchmark model."""
    from trinerflet_amd import synthetic
    poses = synthetic.hemisphere_poses(images, seed=3)
    model.eval()
    model.encoder.reset_cahce()
    times, wide = [], []
    with torch.no_grad():
        for k in range(images + 1):
            pix = np.stack([np.full(640000, k % images, np.int64), np.arange(640000)], -1)
            o, d = synthetic.get_rays(poses, pix)
            o, d = torch.from_numpy(o).to(device)[None], torch.from_numpy(d).to(device)[None]
            for min_step, acc in ((1, times), (8, wide)):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                model.render(o, d, staged=True, bg_color=0, perturb=False, max_steps=max_steps, infer_min_step=min_step)
                torch.cuda.synchronize()
                acc.append(time.perf_counter() - t0)
    model.train()
    t, tw = float(np.mean(times[1:])), float(np.mean(wide[1:]))
    return {"image": "800x800", "max_steps": max_steps, "ms_per_image": round(t * 1e3, 2), "rays_per_s": 640000 / t,
            "wide_iterations": {"ms_per_image": round(tw * 1e3, 2), "rays_per_s": 640000 / tw,
                                "note": "render(..., infer_min_step=8): the same per-ray sample sequences in an eighth of "
                                        "the iterations; identical pixels for rays that end before the max_steps cap "
                                        "(tests/test_renderer_gpu.py)"},
            "note": "run_cuda eval branch (device-driven alive-ray loop, the reference's schedule), solid-sphere "
                    "occupancy, the benchmark's field after its training steps; mean of 3 images after one warm-up"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="base", choices=sorted(WORKLOADS))
    ap.add_argument("--dist-mode", default="sharded", choices=["sharded", "allreduce"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the measured configuration); gloo only to dry-run the N>1 code on one GPU")
    ap.add_argument("--same-device", action="store_true", help="all ranks on cuda:0 (dry-run with --backend gloo)")
    ap.add_argument("--sections", action="store_true", help="print a per-section time breakdown to stderr")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank its own 60 000 rays per step; strong: the 60 000 rays of a step split over the ranks")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary figures (whole-plane / fp32-plane step times, inference, live PMC traffic)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # a rocprofv3 --pmc pass of pmc_traffic_live
    args = ap.parse_args()
    if args.pmc_child:
        args.no_extras = args.no_cpu_baseline = True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from trinerflet_amd import build as tbuild
    from trinerflet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        if rank == 0:
            tbuild.build()
        if world > 1:
            dist.barrier()

    if os.environ.get("TNL_MAIN_PRIO"):    # experiments: the step on a high-priority stream, side work on a normal one
        torch.cuda.set_stream(torch.cuda.Stream(priority=int(os.environ["TNL_MAIN_PRIO"])))
    model, ts, bitfield, N = build(args.workload, device, args.dist_mode if world > 1 else None)
    if os.environ.get("TNL_CLIP_FAR") == "1":      # A/B: march only to the exit from the occupied box (TrainStep.clip_far)
        ts.clip_far = True
    n_global = N * world
    if args.scaling == "strong":       # the step's 60 000 rays split over the ranks (the dense work is what shards)
        n_global = N
        N = N // world
    batches = make_batches(4, N, rank, device)

    # dry run: fixes the per-step sample budget (mean_count) so M is constant
    model.mean_count = 0
    counts = []
    for b in batches:
        one_step(model, ts, bitfield, b, 0)
        counts.append(int(ts.last["counter"][0].item()))
    mean_count = int(max(counts) * 1.02)
    if world > 1:
        mc = torch.tensor([mean_count], device=device)
        dist.all_reduce(mc, op=dist.ReduceOp.MAX)
        mean_count = int(mc.item())
    model.mean_count = mean_count

    nb = len(batches)
    # Set-up, not measurement: one whole density-grid period (16 steps, so the second refresh with its first-use code
    # paths -- sample-budget update, partial ROI rebuild -- has happened once), then park the long-lived Python objects
    # in the permanent generation so that a generation-2 collection (10-20 ms with the torch module tree) cannot land
    # inside the timed steps.  The W warm-up steps and the K timed steps follow unchanged; the timed window still
    # contains its grid refresh (every 16th step).
    for i in range(16):
        one_step(model, ts, bitfield, batches[i % nb], mean_count, batches[(i + 1) % nb])
    torch.cuda.synchronize()
    gc.collect()
    gc.freeze()
    for i in range(args.warmup):
        one_step(model, ts, bitfield, batches[i % nb], mean_count, batches[(i + 1) % nb])

    # HIP events on the launch stream inside the timed steps: only the two boundaries around the dominant kernel's
    # launches (every recorded boundary costs the stream 6-8 us; all of them together were 1.3 % of a step)
    ts.section_events = []
    ts.section_names = {"idwt_adjoint", "scaler_probe", "adam_coef"}
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    flushes0, deferred0 = ts.deferred_flushes, ts.deferred_steps
    for i in range(args.steps):
        one_step(model, ts, bitfield, batches[(args.warmup + i) % nb], mean_count, batches[(args.warmup + i + 1) % nb])
    # the optimiser work TrainStep deferred during these steps (coefficients outside the occupancy window's footprint,
    # replayed in one pass per flush) belongs to them: it is done before the clock stops
    ts.flush_deferred()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    timed_flushes, timed_deferred = ts.deferred_flushes - flushes0, ts.deferred_steps - deferred0
    # the dominant kernel's time from the HIP events recorded inside the timed region ...
    adam_ms = ts.section_times().get("adam_coef", float("nan"))
    # ... and every section's, for the secondary figures, from an instrumented pass AFTER it (not part of the K steps)
    ts.section_events, ts.section_names = [], None
    for i in range(min(args.steps, 16)):
        j = args.warmup + args.steps + i
        one_step(model, ts, bitfield, batches[j % nb], mean_count, batches[(j + 1) % nb])
    torch.cuda.synchronize()
    sec = ts.section_times()
    sec["adam_coef"] = adam_ms
    # If the step starts the next batch's march + tile sort together with the Adam launches (TrainStep.prefetch_at =
    # "adam"), the figure above is the kernel sharing the GPU with them; then also the kernel by itself: the same steps
    # with the side work started after the field backward instead (nothing runs beside Adam).
    adam_alone_ms = float("nan")
    if ts._prefetch_under_adam(((),)):
        ts.section_events, ts.section_names = [], {"idwt_adjoint", "scaler_probe", "adam_coef"}
        ts.prefetch_at = "bwd"
        for i in range(min(args.steps, 16)):
            j = args.warmup + args.steps + 16 + i
            one_step(model, ts, bitfield, batches[j % nb], mean_count, batches[(j + 1) % nb])
        torch.cuda.synchronize()
        adam_alone_ms = ts.section_times().get("adam_coef", float("nan"))
        ts.prefetch_at = os.environ.get("TNL_PREFETCH_AT", "auto")
    samples_per_step = float(np.mean(counts))
    P_coef = ts.coef_numel if ts.dist_mode != "sharded" else ts.coef_numel // world
    # algorithmic bytes of the step's Adam launches (one per wavelet level + LL): 16 B read (p, g, m, v) + 12 B
    # written (p, m, v) per coefficient; with the gradient-support chain g is neither stored nor read outside each
    # level's rectangle (24 B there).  TrainStep._rects holds the rectangles of the last non-refresh step.
    S_own = (3 * ts.C) // (world if ts.dist_mode == "sharded" else 1)
    rects = ts._rects if (ts._rect_ok and ts._roi is not None and all(r is not None for r in ts._rects)) else None
    # with the live / deferred split (TrainStep.defer_adam) a level's per-step launch covers only its live rectangle;
    # the coefficients outside it are replayed by k_adam_l1_catchup once per flush (24 B each, reported separately)
    live = ts.last_live if (ts.defer_adam and rects is not None and ts.last_live is not None) else [None] * ts.J
    adam_bytes, n_launch, deferred_coefs = 0.0, 0, 0.0
    for lvl in range(ts.J):
        n_l = ts.coef.params[lvl].shape[-1]
        inside = rects[lvl][6] * rects[lvl][7] if rects is not None else n_l * n_l
        domain = live[lvl][6] * live[lvl][7] if live[lvl] is not None else n_l * n_l
        adam_bytes += S_own * 3 * (24.0 * domain + 4.0 * inside)
        deferred_coefs += S_own * 3 * float(n_l * n_l - domain)
        n_launch += 1
    n_ll = ts.ll.params[0].shape[-1]
    adam_bytes += S_own * (24.0 * n_ll * n_ll + 4.0 * (rects[0][6] * rects[0][7] if rects is not None else n_ll * n_ll))
    n_launch += 1
    if ts.defer_adam and rects is not None:
        n_launch = 2          # all wavelet levels in one k_adam_l1_live launch + the LL launch
    achieved = adam_bytes / (adam_ms * 1e-3) / 1e9 if adam_ms == adam_ms and adam_ms > 0 else float("nan")
    adam_deferred = None
    if any(lv is not None for lv in live):
        cu_ms = sec.get("adam_catchup", float("nan"))
        all_coefs = float(S_own * 3 * sum(ts.coef.params[lvl].shape[-1] ** 2 for lvl in range(ts.J)))
        adam_deferred = {
            "live_rectangles": [None if lv is None else {"level_size": ts.coef.params[k].shape[-1], "origin_x": lv[0:3],
                                                         "origin_y": lv[3:6], "width": lv[6], "height": lv[7]}
                                for k, lv in enumerate(live)],
            "deferred_share_of_coefficients": round(deferred_coefs / all_coefs, 4),
            "steps_deferred_in_timed_region": timed_deferred, "flushes_in_timed_region": timed_flushes,
            "catchup": None if cu_ms != cu_ms else {
                "ms": round(cu_ms, 4), "records": ts.last_flush_records, "bytes": 24.0 * deferred_coefs,
                "GB/s": round(24.0 * deferred_coefs / (cu_ms * 1e-3) / 1e9, 1)},
            "note": "coefficients outside a level's live rectangle (what the windowed plane rebuild reads + where the "
                    "windowed adjoint writes) are neither read nor reached by a data gradient until the occupancy window "
                    "changes; their Adam(+L1) steps are replayed in registers by k_adam_l1_catchup, all pending steps "
                    "in one 24-B/coefficient pass (bit-identical p, m, v).  Every replay the timed steps caused runs "
                    "inside the timed region (the ring holds 16 steps; a flush also precedes the clock's stop)."}

    # secondary rooflines (north star: HBM GB/s of the sampling / IDWT kernels, MFMA rate of the MLP), from the same
    # HIP-event sections; bytes and flops are the algorithmic ones of SURVEY.md 8(d) for what each section moves
    Cc, Rr, Hh = ts.C, ts.R, ts.H
    Ms = samples_per_step
    e_pl = 2 if model.encoder.plane_dtype == torch.float16 else 4
    mac = 3 * Cc * Hh + 16 * Hh + 31 * Hh + Hh * Hh + 3 * Hh
    wins = ts._forward_windows() if ts._roi is not None else [None] * ts.J

    def win_area(lvl, m):           # texels of level lvl's output that are computed
        w = wins[lvl]
        return float(w[6] * w[7]) if w is not None else float(m * m)
    fwd_bytes = adj_bytes = 0.0
    for lvl in range(ts.J):
        m = Rr >> (ts.J - 1 - lvl)
        out_b = e_pl if lvl == ts.J - 1 else 4
        fwd_bytes += S_own * win_area(lvl, m) * (4.0 + out_b)      # 4 input bands at a quarter of the area + output
        rect = rects[lvl][6] * rects[lvl][7] if rects is not None else (m // 2) ** 2
        adj_bytes += S_own * (win_area(lvl, m) * 4.0 + 4.0 * rect * 4.0)   # gradient window in, 4 bands out
    fwd_bytes += (3 * Cc // (world if ts.dist_mode == "sharded" else 1)) * win_area(ts.J - 1, Rr) * 2 * e_pl  # layout

    def rate(nbytes, key):
        t = sec.get(key, float("nan"))
        return round(nbytes / (t * 1e-3) / 1e9, 1) if t == t and t > 0 else None
    kernels = {
        "field_fwd": {"GB/s": rate(Ms * (12 * Cc * e_pl + 48 + 6 * Cc), "field_fwd"),
                      "mfma_TFLOP/s": rate(2.0 * mac * Ms / 1e3, "field_fwd"),
                      "note": "gather 12*C*e + 48 B/sample + 6*C B/sample of saved features; 2*MAC flops/sample"},
        "field_bwd": {"GB/s": rate(Ms * (6 * Cc + 6 * Cc + 40), "field_bwd"),
                      "mfma_TFLOP/s": rate(6.0 * mac * Ms / 1e3, "field_bwd"),
                      "note": "features in, fp16 dF out; recompute + dX + dW = 6*MAC flops/sample; "
                              "dense fp16 MFMA peak 2500 TFLOP/s"},
        "plane_grad_reduce": {"GB/s": rate(Ms * 3 * (2 * Cc + 12 + 4.4) + 3 * Cc * win_area(ts.J - 1, Rr) * 4, "plane_grad_binned")},
        "idwt_forward_all_levels_plus_layout": {"GB/s": rate(fwd_bytes, "idwt_fwd")},
        "idwt_adjoint_all_levels": {"GB/s": rate(adj_bytes, "idwt_adjoint")},
    }

    # ---- secondary figures, after the timed region (rank 0's GPU only; skipped by --no-extras and in multi-GPU runs)
    extras = {}
    roi_window = None if ts._roi is None else {"origin_x": ts._roi[0:3], "origin_y": ts._roi[3:6], "width": ts._roi[6],
                                                "height": ts._roi[7], "of": ts.R,
                                                "note": "bounding window of the occupied cells per plane (r = 0.8 sphere): "
                                                        "the finest IDWT level, layout change, plane gradient and adjoint "
                                                        "touch only this window on the 15 of 16 steps without a grid refresh"}
    traffic, traffic_src = pmc_traffic_static(args.workload, world)
    if world == 1 and not args.no_extras:
        extras["inference"] = inference_figure(model, device)
        live, note = pmc_traffic_live(args.workload, n_launch)
        if live is not None:
            traffic, traffic_src = live, note
        elif traffic_src is not None:
            traffic_src += f"; live collection unavailable ({note})"
        plc = ts.placement
        del ts, model
        gc.collect()
        torch.cuda.empty_cache()
        k = min(args.steps, 12)
        extras["no_roi_ms_per_step"] = round(variant_ms(args.workload, device, None, batches, mean_count, k, use_roi=False), 4)
        extras["fp32_planes_ms_per_step"] = round(variant_ms(args.workload, device, None, batches, mean_count, k,
                                                             plane_dtype=torch.float32), 4)
        # the other two README configurations at their full geometry: one density-grid period (16 steps, its refresh and the
        # replay of the deferred optimiser pass inside) after a 17-step set-up, same rays and sample budget
        others = {}
        for wl in ("small", "large"):
            if wl == args.workload:
                continue
            t_ms = variant_ms(wl, device, None, batches, mean_count, 16)
            others[wl] = {"ms_per_step": round(t_ms, 4), "rays_per_s": N / (t_ms * 1e-3),
                          "config": "3x{}ch x {}^2, scale {}, hidden {}".format(*WORKLOADS[wl][:4])}
        extras["other_workloads"] = others
        extras["variants_note"] = ("no_roi: every step rebuilds / differentiates whole planes (no occupancy window, no "
                                   "gradient-support rectangles); fp32_planes: the sampler reads fp32 planes as the "
                                   "reference's training does (SURVEY F9; implies whole planes); same rays, budget, "
                                   f"{k} steps without a grid refresh after a 17-step set-up")
    else:
        plc = ts.placement

    if rank == 0:
        C, R, scale, H, _, lam = WORKLOADS[args.workload]
        ms = elapsed / args.steps * 1e3
        survey_bytes = (44 + 2) * 3.0 * C * R * R + samples_per_step * (12 * C * 2 + 48 * C + 64) + 64.0 * N
        wire = None
        if world > 1:
            S_all = 3 * C
            if args.dist_mode == "sharded":
                win = (roi_window["width"] * roi_window["height"]) if roi_window else R * R
                rs = S_all * win * 4.0 * (world - 1) / world       # reduce-scatter of the fp32 plane-gradient window
                ag = S_all * win * 2.0 * (world - 1) / world       # all-gather of the rebuilt fp16 planes (window)
                wire = {"reduce_scatter_plane_grad_bytes": rs, "all_gather_planes_bytes": ag,
                        "note": "sent per rank and step without a grid refresh (whole planes on refresh steps); "
                                "plus 54 kB of MLP-gradient all-reduce and three scalars"}
            else:
                wire = {"all_reduce_plane_grad_bytes": 2.0 * S_all * R * R * 4.0 * (world - 1) / world}
        out = {
            "metric": "train rays/sec (whole node)", "value": N * world * args.steps / elapsed, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: 3x{C}ch x {R}^2 planes, bior6.8 scale {scale}, hidden {H}, "
                                   f"{N} rays/step/GPU, solid-sphere occupancy r=0.8 (re-imposed after each refresh), "
                                   f"fp16 planes + fp16 MFMA MLP, fp32 masters, Adam+L1",
                       "rays_per_step_per_gpu": N, "rays_per_step_global": n_global,
                       "samples_per_step_per_gpu": samples_per_step,
                       "sample_budget_M": mean_count, "parallelism": f"ray-dp{world}" + (f"+{args.dist_mode}" if world > 1 else ""),
                       "collectives": None if world == 1 else {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                                               "bytes_on_the_wire": wire},
                       "samples_per_sec": samples_per_step * world * args.steps / elapsed,
                       # SURVEY.md 8(d)'s byte count of the REFERENCE's step (every coefficient rebuilt, differentiated and
                       # updated every step: (44 + e) P + M (12 C e + 48 C + 64) + 64 N) over this step's time
                       "survey_step": {"bytes": survey_bytes, "TB/s_equivalent": round(survey_bytes / (ms * 1e-3) / 1e12, 3),
                                       "of_8_TB/s": round(survey_bytes / (ms * 1e-3) / 8e12, 3),
                                       "note": "bytes the reference's formulation of one step moves per GPU (SURVEY.md "
                                               "8(d), e = 2) divided by the measured step time; the step itself moves "
                                               "fewer (occupancy window, live rectangles of the optimiser pass)"},
                       "sections_ms": {k: round(v, 4) for k, v in sec.items()},
                       "sections_note": "adam_coef: HIP events inside the timed steps; the other sections: an "
                                        "instrumented pass after them (an event at every boundary costs 6-8 us)",
                       "kernels": kernels,
                       "roi_window": roi_window,
                       "adam_placement": plc,
                       "adam_deferred": adam_deferred,
                       **extras},
            "roofline": {"bound": "hbm", "kernel": f"k_adam_l1 / k_adam_l1_live (fused Adam + wavelet-L1): the step's {n_launch} "
                                                   "launches (all wavelet levels in one launch restricted to their live "
                                                   "rectangles + LL; without the deferral one launch per level), "
                                                   "byte-weighted",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if traffic is None else traffic / n_launch, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": adam_bytes / n_launch,
                         "avg_launch_ms": adam_ms / n_launch, "launches_per_step": n_launch,
                         "per_step": {"algorithmic_bytes": adam_bytes, "traffic_bytes": traffic, "ms": adam_ms},
                         "alone": None if adam_alone_ms != adam_alone_ms else {
                             "ms_per_step": adam_alone_ms, "achieved": adam_bytes / (adam_alone_ms * 1e-3) / 1e9,
                             "frac": adam_bytes / (adam_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "note": "the same launches with nothing beside them (only reported when the step is "
                                     "configured to run the next batch's march + tile sort underneath the Adam pass)"},
                         "note": "achieved = algorithmic bytes of the step's k_adam_l1 launches / their summed duration "
                                 "(HIP events on the launch stream, inside the timed steps) = mean bytes per launch / mean "
                                 "launch duration; 28 B per coefficient inside a level's gradient-support rectangle, 24 B "
                                 "outside it (g = 0 is neither stored nor read there), nothing outside a live rectangle "
                                 "(config.adam_deferred); 8000 GB/s is the spec peak, a "
                                 "float4 copy reaches 6290 GB/s on MI355X (MI355X_MICROARCH.md)"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        if args.sections:
            print(json.dumps(sec, indent=1), file=sys.stderr)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
