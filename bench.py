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

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` -- the section of the step with the
largest per-step time (found by an instrumented pass before the timed steps, then timed with HIP events inside them),
its binding roof (HBM bytes or MFMA flops, whichever takes longer at peak), `roofline.top` = the three longest
sections with algorithmic bytes / flops, fraction and counter traffic, the step's actual traffic, this box's measured
copy bandwidth -- and `cpu_baseline`.  `config.trajectory`: 512 steps from an untrained grid (tools/trajectory.py).
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
MFMA_PEAK_TFLOPS = 2500.0  # dense fp16 MFMA peak (MI355X_MICROARCH.md; the headline figures with 2:1 sparsity are not used)

# section -> the mark that precedes it in TrainStep.step (a section is timed between two consecutive marks)
SECTION_PREV = {"idwt_fwd": "begin", "field_fwd": "march", "field_bwd": "composite_bwd", "plane_grad_binned": "field_bwd",
                "idwt_adjoint": "scaler_probe", "adam_coef": "idwt_adjoint"}
# section -> substrings of the kernel names launched inside it (for the rocprofv3 counter passes and profiles/)
SECTION_KERNELS = {"field_fwd": ["k_field_fwd"], "field_bwd": ["k_field_bwd", "k_slab_reduce"],
                   "adam_coef": ["k_adam_l1_live", "k_adam_l1", "k_adam_record"],
                   "plane_grad_binned": ["k_tile_accumulate"], "idwt_fwd": ["k_idwt_fwd", "k_to_texel_major"],
                   # (k_idwt_bwd_walk<W, true>: the column-walk levels with the optimiser's live pass in their epilogue, TrainStep.fuse_live)
                   "idwt_adjoint": ["k_idwt_bwd_walk", "k_idwt_bwd_pipe"]}

def section_owns(section, kernel):
    """Whether a kernel (its short name, _short) belongs to a section: a listed name or a listed name + "_suffix"
    (k_field_bwd -> k_field_bwd_rows, k_idwt_bwd_walk -> k_idwt_bwd_walk_..., k_to_texel_major -> k_to_texel_major_h).
    The replay of the deferred optimiser pass (k_adam_l1_catchup) is its own section, not part of the per-step pass."""
    if kernel.startswith("k_adam_l1_catchup"):
        return False
    return any(kernel == sub or kernel.startswith(sub + "_") for sub in SECTION_KERNELS[section])


WORKLOADS = {
    # name: (channels, resolution, wavelet scale, hidden, rays, lambda)
    "base": (32, 2048, 32, 64, 60000, 0.4),
    "small": (16, 1024, 16, 64, 60000, 0.2),
    "large": (48, 2048, 32, 128, 60000, 0.6),
    "tiny": (16, 256, 4, 64, 4096, 0.2),
}


GRAPH = False   # TrainStep(graph=...) for every model this run builds: set in main() (--graph / --no-graph)


def build(workload, device, dist_mode, plane_dtype=None, shell=(0.8, 0.0), **ts_kwargs):
    from trinerflet_amd import synthetic
    ts_kwargs.setdefault("graph", GRAPH and dist_mode is None)
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
    if os.environ.get("TNL_LIVE_BANDS"):      # A/B: 0 = whole live rectangles
        ts.live_bands = os.environ["TNL_LIVE_BANDS"] != "0"
    if os.environ.get("TNL_EXCHANGE_BANDS"):  # the plane-gradient window reduced / exchanged in this many bands of rows
        ts.overlap_exchange = int(os.environ["TNL_EXCHANGE_BANDS"])
    if os.environ.get("TNL_SIDE_COUNT_FORM"):  # A/B: count pass of the prefetched march (0 wavefront per ray, 1 ray per lane)
        ts.side_count_form = int(os.environ["TNL_SIDE_COUNT_FORM"])
    if os.environ.get("TNL_PREFETCH_AT"):     # A/B: where the next batch's march + tile sort start (bwd | reduce | adjoint)
        ts.prefetch_at = os.environ["TNL_PREFETCH_AT"]
    if os.environ.get("TNL_SIDE_CAPS"):       # A/B: "emit,fill" workgroups of the prefetched march's wide passes (0 = uncapped)
        ts.side_caps = tuple(int(x) for x in os.environ["TNL_SIDE_CAPS"].split(","))
    if os.environ.get("TNL_IDWT_TUNING"):     # A/B: "key=value,..." of tnl_idwt_set_tuning (3: rows per walk workgroup, 4: forward form)
        from trinerflet_amd import _lib as _L
        for kv in os.environ["TNL_IDWT_TUNING"].split(","):
            k_, v_ = kv.split("=")
            assert _L.lib().tnl_idwt_set_tuning(int(k_), int(v_)) == 0
    if os.environ.get("TNL_FUSE_LIVE"):       # A/B: 0 = separate adjoint and optimiser passes on every level
        ts.fuse_live = os.environ["TNL_FUSE_LIVE"] != "0"
        ts.fuse_live_levels = max(int(os.environ["TNL_FUSE_LIVE"]), 1)
    if os.environ.get("TNL_NO_OVERLAP"):      # experiments: march + tile sort in order on the launch stream (kernels alone)
        ts.overlap_march = False
    bitfield = torch.from_numpy(synthetic.sphere_bitfield(128, model.cascade, 1.5, shell[0], shell[1])).to(device)
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
    """Reference operator set on the host cores (oracle/torch_baseline.py), BASELINE.md section 3's procedure: the dense
    part (plane rebuild, its backward, regulariser, Adam) at the FULL plane size, the per-ray part on a reduced ray count
    scaled linearly; one warm-up step, then the median of 5 timed steps (section 3: >= 5)."""
    from oracle import torch_baseline as tb
    C, R, scale, H, N, lam = WORKLOADS[workload]
    # torch's CPU kernels stop scaling (and then regress) well before the 256 hardware threads of the GPU box's
    # host: 32 threads measured fastest there (8: 1.38 s, 16: 1.23 s, 32: 1.10 s, 64: 1.92 s for the same sample)
    cores = min(os.cpu_count() or 1, 32)
    Ns = max(N // 20, 64)
    t = tb.time_step(C, R, scale, H, Ns, lam=lam, threads=cores, repeats=5, warmup=1)
    dense, ray = t["dense_s"], t["ray_s"] * (N / Ns)
    return {"value": N / (dense + ray), "unit": "rays/s", "cores": cores, "host_threads": os.cpu_count(), "kind": "port",
            "sample_short": f"torch-CPU fp32 step on {cores} threads of {os.cpu_count()}: dense part (IDWT fwd+bwd, L1, Adam) at full "
                            f"R={R}, per-ray part on {Ns} of {N} rays scaled x{N / Ns:.0f}; 1 warm-up + median of 5",
            "samples_per_s": N * 512 / (dense + ray),
            "split_s": {"dense_full_size": round(dense, 3), "per_ray_scaled": round(ray, 3)},
            "sample": f"torch-CPU fp32 step, 1 warm-up + median of 5: dense part (IDWT fwd+bwd, L1, Adam over "
                      f"{3 * C * R * R / 1e6:.0f} M coefficients) at the full R={R}: {[round(v, 2) for v in t['dense_all_s']]} s; "
                      f"per-ray part (512 uniform steps/ray, renderer.run semantics, on the full-size planes) on {Ns} rays "
                      f"scaled x{N / Ns:.0f}: {[round(v, 2) for v in t['ray_all_s']]} s"}


def _short(name):
    import re
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:48]


def pmc_traffic_live(workload):
    """HBM traffic counters of EVERY kernel of a steady-state step, measured NOW: two child runs of this script under
    `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE in separate passes as the guide prescribes, no trace domain, the
    program itself after `--`).  The window is the child's last two steps (no grid refresh inside), delimited by the
    k_step_epilogue dispatches.  Returns ({kernel: {"FETCH_SIZE": KB per step, "WRITE_SIZE": KB per step}}, note) or
    (None, reason).  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of a wide
    coalesced read, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; the raw sum is reported beside it."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    per = {}
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
            rows = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] == counter:
                            rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], float(row["Counter_Value"])))
            rows.sort()
            ends = [i for i, kn, _ in rows if "k_step_epilogue" in kn]
            if len(ends) < 3:
                return None, f"no step boundaries in the {counter} pass"
            lo, hi = ends[-3], ends[-1]
            for i, kn, v in rows:
                if lo < i <= hi:
                    per.setdefault(_short(kn), {}).setdefault(counter, 0.0)
                    per[_short(kn)][counter] += v / 2.0
    return per, ("live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run (separate passes), the child's last "
                 "two steps; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per the guide's gfx950 correction")


def copy_bandwidth(device, nbytes=1 << 30, reps=10):
    """What this box's memory system gives a plain streaming copy (16 B per lane, tnl_copy_probe): GB/s of read + write."""
    from trinerflet_amd import _lib as L
    a = torch.empty(nbytes, dtype=torch.uint8, device=device)
    b = torch.empty_like(a)
    run = lambda: L.check(L.lib().tnl_copy_probe(L.ptr(a), L.ptr(b), L.u64(nbytes), L.stream()), "copy_probe")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    e1.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


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


def variant_report(workload, device, batches, steps=16, shell=(0.8, 0.0), **kw):
    """A second configuration measured like the headline one, in short: its own dry run for the sample budget, a
    17-step set-up (past the refresh at step 16), `steps` timed steps (one whole density-grid period at 16: its refresh and
    the replay of the deferred pass inside), then an instrumented period for the sections -> ms per step, the longest
    section and its roofline fraction (same algorithmic-byte table as the headline)."""
    model, ts, bitfield, N = build(workload, device, None, shell=shell, **kw)
    model.mean_count = 0
    counts = []
    for b in batches:
        one_step(model, ts, bitfield, b, 0)
        counts.append(int(ts.last["counter"][0].item()))
    mean_count = int(max(counts) * 1.02)
    model.mean_count = mean_count
    t_ms = time_steps(model, ts, bitfield, batches, mean_count, steps, setup=17) * 1e3
    ts.section_events, ts.section_names = [], None
    nb = len(batches)
    for i in range(16):
        one_step(model, ts, bitfield, batches[i % nb], mean_count, batches[(i + 1) % nb])
    torch.cuda.synchronize()
    sec = ts.section_times("median")
    ts.section_events = None
    acct = adam_accounting(ts, 1)
    spec, launches = section_specs(ts, model, float(np.mean(counts)), 1, acct)
    dominant = max(SECTION_PREV, key=lambda k: sec.get(k, 0.0))
    rf = roof(spec, dominant, sec.get(dominant, float("nan"))) or {}
    C, R, scale, H = WORKLOADS[workload][:4]
    out = {"ms_per_step": round(t_ms, 4), "rays_per_s": N / (t_ms * 1e-3), "samples_per_step": float(np.mean(counts)),
           "config": f"3x{C}ch x {R}^2, scale {scale}, hidden {H}" + ("" if shell == (0.8, 0.0) else f", occupancy r in [{shell[1]}, {shell[0]}]"),
           "sections_ms_median": {k: round(v, 4) for k, v in sec.items()},
           "roofline": {"section": dominant, "kernels": SECTION_KERNELS[dominant], "bound": rf.get("bound"),
                        "achieved": rf.get("achieved"), "peak": rf.get("peak"), "unit": rf.get("unit"), "frac": rf.get("frac"),
                        "ms_per_step": rf.get("ms_per_step"), "launches_per_step": launches[dominant],
                        "algorithmic_bytes": rf.get("algorithmic_bytes"), "algorithmic_flops": rf.get("algorithmic_flops"),
                        "note": "the section with the largest median time over an instrumented density-grid period; "
                                "traffic counters: profiles/r06_<workload>_* (tools/profile_round.sh --workload)"},
           "sections_roofline": {k: (lambda r_: None if r_ is None else {"ms": r_["ms_per_step"], "bound": r_["bound"],
                                                                         "frac": round(r_["frac"], 4)})(roof(spec, k, sec.get(k, float("nan"))))
                                 for k in spec}}
    del model, ts
    gc.collect()
    torch.cuda.empty_cache()
    return out


def inference_figure(model, device, max_steps=4096, images=3):
    """BASELINE config 5's `--test` render: 800 x 800 rays of one pose through run_cuda's inference branch at
    max_steps = 4096 (renderer.py:324-374), the trained-state planes of the benchmark model.  Three forms: the
    one-kernel render (csrc/render.hip, the default of the eval branch), the device-driven alive-ray loop with the
    reference's schedule, and that loop with 8-sample iterations.  SURVEY.md 8(d) "Inference unit":
    bytes = samples * (12*C*e + 48) + sum over iterations of n_alive * (4 + 4 + 20*2) for the loop; the one-kernel
    render moves the gathers only (samples * 12*C*e) plus 52 B per ray."""
    from trinerflet_amd import synthetic
    import trinerflet_amd.raymarching as rm
    poses = synthetic.hemisphere_poses(images, seed=3)
    model.eval()
    model.encoder.reset_cahce()
    modes = {"kernel": {}, "loop": {"device_loop": True}, "loop_wide": {"infer_min_step": 8}}
    times = {k: [] for k in modes}
    sched = []
    with torch.no_grad():
        rays = []
        for k in range(images + 1):
            pix = np.stack([np.full(640000, k % images, np.int64), np.arange(640000)], -1)
            o, d = synthetic.get_rays(poses, pix)
            rays.append((torch.from_numpy(o).to(device)[None], torch.from_numpy(d).to(device)[None]))
        # each form renders its frames back to back, as `--test` renders a sequence of poses with one form (the first frame of
        # a form is its warm-up); interleaved per frame with the two loop forms the one-kernel render took 5.45 / 6.09 ms against 5.03 / 5.56
        for name, kw in modes.items():
            for o, d in rays:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                model.render(o, d, staged=True, bg_color=0, perturb=False, max_steps=max_steps, **kw)
                torch.cuda.synchronize()
                times[name].append(time.perf_counter() - t0)
        for k in (images,):
            o, d = rays[k]
            if k == images:
                # the schedule of that image (iterations, survivors, samples): the host-driven loop runs the same
                # iterations (bit-identical image, tests/test_renderer_gpu.py) and passes them through march_rays
                real = rm.march_rays

                def logging_march(n_alive, n_step, *a, **kw):
                    sched.append((int(n_alive), int(n_step)))
                    return real(n_alive, n_step, *a, **kw)
                rm.march_rays = logging_march
                try:
                    out = model.render(o, d, staged=True, bg_color=0, perturb=False, max_steps=max_steps, device_loop=False)
                finally:
                    rm.march_rays = real
    model.train()
    t = {k: float(np.mean(v[1:])) for k, v in times.items()}
    C = model.encoder.number_of_features
    e = 2 if model.encoder.plane_dtype == torch.float16 else 4
    slots = float(sum(a * b for a, b in sched))          # sample rows the loop evaluates (padding of dead rays included)
    alive = float(sum(a for a, _ in sched))
    loop_bytes = slots * (12 * C * e + 48) + alive * 48.0
    H = model.hidden_dim
    mac = 3 * C * H + 16 * H + 31 * H + H * H + 3 * H

    def fig(seconds, nbytes, samples):
        return {"ms_per_image": round(seconds * 1e3, 2), "rays_per_s": 640000 / seconds, "samples_per_s": samples / seconds,
                "algorithmic_bytes": nbytes, "GB/s": round(nbytes / seconds / 1e9, 1),
                "frac_of_8_TB/s": round(nbytes / seconds / 8e12, 4),
                "mlp_TFLOP/s": round(2.0 * mac * samples / seconds / 1e12, 1)}
    kernel_bytes = slots * 12 * C * e + 640000 * 52.0
    res = {"image": "800x800", "max_steps": max_steps, **fig(t["kernel"], kernel_bytes, slots),
           "form": "one persistent kernel (tnl_render_rays): march + fused field + compositing per ray, rays from a queue, "
                   "no sample buffers; bytes = the texel gathers (samples * 12*C*e) + 52 B per ray",
           "loop": {**fig(t["loop"], loop_bytes, slots), "iterations": len(sched), "sum_n_alive_over_iterations": alive,
                    "note": "the device-driven alive-ray loop with the reference's schedule (render(..., device_loop=True)); "
                            "bytes = SURVEY.md 8(d) inference unit"},
           "loop_wide_iterations": {**fig(t["loop_wide"], loop_bytes, slots),
                                    "note": "render(..., infer_min_step=8): the same per-ray sample sequences in an eighth "
                                            "of the iterations"},
           "samples_per_image": slots,
           "note": "run_cuda eval branch, solid-sphere occupancy, the benchmark's field after its training steps (a "
                   "transparent random field: few rays end by transmittance); mean of 3 images after one warm-up; "
                   "rocprofv3 summaries: profiles/r03c_infer_kernel_stats.csv"}
    return res


def dropin_figure(workload, device, batches, mean_count):
    import importlib.util
    spec = importlib.util.spec_from_file_location("tnl_bench_dropin", os.path.join(ROOT, "tools", "bench_dropin.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    out = {}
    for name, (opt, planes) in {"fused_adam_fp16_planes": ("fused", "fp16"), "fused_adam_fp32_planes": ("fused", "fp32"),
                                "torch_adam_fp32_planes": ("adam", "fp32")}.items():
        out[name] = round(D.dropin_ms_per_step(workload, device, batches, mean_count, 32, opt, planes), 3)
    out["note"] = ("the loop of reconstruction/nerf/utils.py:1134-1175 (get_planes -> render -> MSE + wavelet L1 through "
                   "autograd -> scaler.scale(loss).backward() -> scaler.step(optimizer)) on the drop-in modules, 32 steps = two "
                   "density-grid periods, same rays / occupancy / sample budget as the headline.  torch_adam_fp32_planes: "
                   "main_nerf.py unchanged (torch.optim.Adam, the reference's fp32 training planes); fused_adam_*: the one-line "
                   "change at main_nerf.py:119 to trinerflet_amd.optim.FusedAdamL1 (INTEGRATION.md); fp16 planes = the "
                   "encoder's default plane_dtype.  Round 3 measured 30.7 ms/step for this loop over a trajectory (24.5 "
                   "steady state): since then the autograd backward goes through the tile-sorted reduction instead of "
                   "global float atomics, the regulariser's |x|.mean() is one fused pass forward and a scalar handed to the "
                   "optimiser backward (FusedAdamL1 fold_l1), Adam one pass per parameter, GradScaler's inf check a "
                   "read-only pass, the sample budget's zero padding is skipped, and (install_dropin()'s "
                   "windowed_autograd, on in these figures) the plane rebuild / layout pass / adjoint cover the occupancy "
                   "window only")
    return out


def trajectory_figure(workload, device, steps=512):
    """config.trajectory: the fused step on a REAL trajectory (tools/trajectory.py): untrained grid -> real refreshes,
    occupancy window and sample budget as they evolve, held-out PSNR at the end."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tnl_trajectory", os.path.join(ROOT, "tools", "trajectory.py"))
    T = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(T)
    if workload not in T.GEOM:
        return None
    rep = T.run_fused(workload, device, steps, WORKLOADS[workload][4])
    model = rep.pop("_model")
    # the `--test` render on the TRAINED field of this trajectory (opaque ball: rays end by transmittance, the regime
    # renderer.py:324-374 sees), beside config.inference's figure on the benchmark's transparent random field
    try:
        inf = inference_figure(model, device)
        rep["inference_trained"] = {k: inf[k] for k in ("image", "max_steps", "ms_per_image", "rays_per_s", "samples_per_s",
                                                         "samples_per_image", "GB/s", "frac_of_8_TB/s", "mlp_TFLOP/s")}
        rep["inference_trained"]["loop_ms_per_image"] = inf["loop"]["ms_per_image"]
        rep["inference_trained"]["note"] = "800x800, max_steps 4096, the field trained by this trajectory (its own occupancy grid)"
    except Exception as e:                                   # noqa: BLE001
        rep["inference_trained"] = {"error": str(e)}
    del model
    periods = rep.pop("periods")
    rep["per_16_steps"] = {"ms_per_step": [p["ms_per_step"] for p in periods],
                           "refresh_step_ms": [p["refresh_step_ms"] for p in periods],
                           "samples_per_step": [p["samples_per_step"] for p in periods],
                           "window": [p["window"] for p in periods]}
    gc.collect()
    torch.cuda.empty_cache()
    return rep


def adam_accounting(ts, world):
    """Algorithmic bytes of the step's coefficient pass and what it leaves to the deferred replay."""
    # algorithmic bytes of the step's Adam launches (one per wavelet level + LL): 16 B read (p, g, m, v) + 12 B
    # written (p, m, v) per coefficient; with the gradient-support chain g is neither stored nor read outside each
    # level's rectangle (24 B there).  TrainStep._rects holds the rectangles of the last non-refresh step.
    S_own = (3 * ts.C) // (world if ts.dist_mode == "sharded" else 1)
    rects = ts._rects if (ts._rect_ok and ts._roi is not None and all(r is not None for r in ts._rects)) else None
    # with the live / deferred split (TrainStep.defer_adam) a level's per-step launch covers only its live rectangle;
    # the coefficients outside it are replayed by k_adam_l1_catchup once per flush (24 B each, reported separately)
    live = ts.last_live if (ts.defer_adam and rects is not None and ts.last_live is not None) else [None] * ts.J
    tables = ts.last_live_bands if (ts.last_live_bands is not None and live[0] is ts.last_live[0]) else [None] * ts.J
    adam_bytes, n_launch, deferred_coefs, fused_bytes = 0.0, 0, 0.0, 0.0
    band_share = [None] * ts.J
    fused = tuple(ts.last_fused_levels) if (ts.fuse_live and ts.defer_adam and rects is not None) else ()

    def band_counts(lv, tbl, r):
        """Coefficients per (slice, band) of a level's band pieces, and how many of them lie inside the stored-gradient
        rectangle r (mean over the planes)."""
        nb = lv[7] // 8
        w = 4 * tbl[nb + 1:2 * nb + 1].astype(np.int64)
        dom, ins = float(8 * w.sum()), 0.0
        for pl in range(3):
            x0 = tbl[2 * nb + 1 + pl * nb:2 * nb + 1 + (pl + 1) * nb].astype(np.int64)
            cols = np.clip(np.minimum(x0 + w, r[pl] + r[6]) - np.maximum(x0, r[pl]), 0, None)
            y0 = lv[3 + pl] + 8 * np.arange(nb)
            rows = np.clip(np.minimum(y0 + 8, r[3 + pl] + r[7]) - np.maximum(y0, r[3 + pl]), 0, None)
            ins += float((cols * rows).sum()) / 3
        return dom, ins
    for lvl in range(ts.J):
        n_l = ts.coef.params[lvl].shape[-1]
        inside = rects[lvl][6] * rects[lvl][7] if rects is not None else n_l * n_l
        domain = live[lvl][6] * live[lvl][7] if live[lvl] is not None else n_l * n_l
        if live[lvl] is not None and tables[lvl] is not None:
            rect_area = domain
            domain, inside = band_counts(live[lvl], tables[lvl][2], rects[lvl])
            band_share[lvl] = round(domain / rect_area, 4)
        if lvl in fused:
            # the adjoint level carries the optimiser (TrainStep.fuse_live): p, m, v read and written, the band gradients
            # never leave the kernel -- counted with the adjoint section
            fused_bytes += S_own * 3 * 24.0 * domain
        else:
            adam_bytes += S_own * 3 * (24.0 * domain + 4.0 * inside)
        deferred_coefs += S_own * 3 * float(n_l * n_l - domain)
        n_launch += 1
    n_ll = ts.ll.params[0].shape[-1]
    adam_bytes += S_own * (24.0 * n_ll * n_ll + 4.0 * (rects[0][6] * rects[0][7] if rects is not None else n_ll * n_ll))
    n_launch += 1
    if ts.defer_adam and rects is not None:
        n_launch = 2          # all wavelet levels in one k_adam_l1_live launch + the LL launch
    return {"adam_bytes": adam_bytes, "n_launch": n_launch, "deferred_coefs": deferred_coefs, "band_share": band_share,
            "live": live, "rects": rects, "S_own": S_own, "fused_levels": fused, "fused_bytes": fused_bytes}


def section_specs(ts, model, samples_per_step, world, acct):
    """Algorithmic bytes / flops per section of one step and the launches of each section's main kernel."""
    S_own, rects, adam_bytes = acct["S_own"], acct["rects"], acct["adam_bytes"]
    # algorithmic bytes / flops per section for what each section moves
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
        if lvl in acct["fused_levels"]:
            # gradient window in, the low-pass band out over the live rectangle; the three detail bands feed the optimiser
            # in the kernel (its 24 B per live coefficient: acct["fused_bytes"], added below)
            adj_bytes += S_own * (win_area(lvl, m) * 4.0 + 4.0 * acct["live"][lvl][6] * acct["live"][lvl][7])
        else:
            adj_bytes += S_own * (win_area(lvl, m) * 4.0 + 4.0 * rect * 4.0)   # gradient window in, 4 bands out
    adj_bytes += acct["fused_bytes"]
    fwd_bytes += (3 * Cc // (world if ts.dist_mode == "sharded" else 1)) * win_area(ts.J - 1, Rr) * 2 * e_pl  # layout
    spec = {
        "field_fwd": {"bytes": Ms * (12 * Cc * e_pl + 48 + 6 * Cc), "flops": 2.0 * mac * Ms,
                      "per_unit": "per sample: gather 12*C*e + 48 B + 6*C B of saved features; 2*MAC flops (MAC = 13 440 at base)"},
        "field_bwd": {"bytes": Ms * (6 * Cc + 6 * Cc + 40), "flops": 6.0 * mac * Ms,
                      "per_unit": "per sample: 6*C B features in + 6*C B fp16 dF out + 40 B (xyz, dir, g_sigma, g_rgb); "
                                  "recompute + dX + dW = 6*MAC flops"},
        "adam_coef": {"bytes": adam_bytes, "flops": 0.0,
                      "per_unit": "per live coefficient of the levels the adjoint has not updated already: 28 B inside the gradient rectangle, 24 B outside it; nothing outside a live rectangle"},
        "plane_grad_binned": {"bytes": Ms * 3 * (2 * Cc + 1.13 * 12) + 3 * Cc * win_area(ts.J - 1, Rr) * 4, "flops": 0.0,
                              "per_unit": "per sample and plane: 2*C B dF + 1.13 list entries of 12 B (sample id + its texel coordinates on the plane); + 4 B per window texel and channel stored"},
        "idwt_fwd": {"bytes": fwd_bytes, "flops": 0.0,
                     "per_unit": "per computed output texel and slice: 4 B of input bands + e (finest) or 4 B out, + 2*e for the texel-major layout pass"},
        "idwt_adjoint": {"bytes": adj_bytes, "flops": 0.0,
                         "per_unit": "per slice: 4 B per gradient-window texel in + 16 B per coefficient-rectangle position out; on the "
                                     "levels that carry the optimiser (fuse_live): 4 B in + 4 B of low-pass gradient out per live-rectangle "
                                     "position + 24 B per live coefficient (p, m, v read and written; the band gradients stay in registers)"},
    }
    launches = {"field_fwd": 1, "field_bwd": 1 if Hh == 64 else 2, "adam_coef": acct["n_launch"], "plane_grad_binned": 1,
                "idwt_fwd": ts.J + 1, "idwt_adjoint": ts.J}
    return spec, launches


def roof(spec, name, ms_):
    """The binding roof of a section: the longer of bytes / HBM peak and flops / MFMA peak."""
    sp = spec[name]
    if not (ms_ == ms_ and ms_ > 0):
        return None
    t_hbm = sp["bytes"] / (HBM_PEAK_GBS * 1e9)
    t_mfma = sp["flops"] / (MFMA_PEAK_TFLOPS * 1e12)
    bound = "mfma" if t_mfma > t_hbm else "hbm"
    gbs = sp["bytes"] / (ms_ * 1e-3) / 1e9
    tfl = sp["flops"] / (ms_ * 1e-3) / 1e12
    out = {"section": name, "kernels": SECTION_KERNELS[name], "ms_per_step": round(ms_, 4), "bound": bound,
           "achieved": tfl if bound == "mfma" else gbs, "peak": MFMA_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBS,
           "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
           "frac": (tfl / MFMA_PEAK_TFLOPS) if bound == "mfma" else (gbs / HBM_PEAK_GBS),
           "algorithmic_bytes": sp["bytes"], "algorithmic_flops": sp["flops"], "per_unit": sp["per_unit"],
           "GB/s": round(gbs, 1), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4)}
    if sp["flops"] > 0:
        out["mfma_TFLOP/s"] = round(tfl, 1)
        out["frac_mfma"] = round(tfl / MFMA_PEAK_TFLOPS, 4)
    return out

def _num(v, nd=4):
    """Scalars of the printed line: floats rounded to `nd` significant-enough digits, everything else unchanged."""
    if isinstance(v, float):
        return float(f"{v:.{nd + 3}g}")
    return v


def compact_line(out):
    """The ONE stdout line of the contract, derived from the full record `out` (which goes to gpurun_out/bench_detail.json):
    the contract's keys, a flat `config` of scalars, `roofline` (scalars + a 3-entry `top` of scalars) and `cpu_baseline`
    (scalars).  Kept below 4 KB so that any driver-side line buffer holds it."""
    cfg, rf = out["config"], out["roofline"]
    flat = {"workload": cfg["workload"]}
    for k in ("rays_per_step_per_gpu", "rays_per_step_global", "samples_per_step_per_gpu", "sample_budget_M", "samples_per_sec",
              "ms_per_step_over_whole_periods", "fp32_planes_ms_per_step", "no_roi_ms_per_step", "thin_shell_ms_per_step",
              "parallelism"):
        if k in cfg:
            flat[k] = _num(cfg[k])
    col = cfg.get("collectives")
    flat["collectives"] = None if col is None else f"{col['backend']} x{col['world_size']}"
    if col is not None and col.get("bytes_on_the_wire"):
        for k, v in col["bytes_on_the_wire"].items():
            if isinstance(v, (int, float)):
                flat["wire_" + k] = _num(float(v))
    ep = cfg.get("exchange_plan")
    if isinstance(ep, dict):     # what the cost model of DESIGN.md section 5 predicts for this run: check ms_per_step against it
        flat["exchange_bands"], flat["exchange_transport"] = ep.get("bands"), ep.get("transport")
        flat["model_ms_per_step"], flat["model_link_GBs"] = ep.get("ms_predicted"), ep.get("link_GBs_assumed")
    tj = cfg.get("trajectory")
    if isinstance(tj, dict):
        flat["trajectory_512_steps_ms_per_step"] = tj.get("wall_ms_per_step")
        flat["trajectory_held_out_psnr_db"] = tj.get("held_out_psnr_db")
        it = tj.get("inference_trained")
        if isinstance(it, dict) and "ms_per_image" in it:      # the `--test` render on the field that trajectory trained
            flat["test_render_trained_800x800_ms"] = it["ms_per_image"]
            flat["test_render_trained_frac_of_8_TB/s"] = it.get("frac_of_8_TB/s")
    inf = cfg.get("inference")
    if isinstance(inf, dict):
        flat["test_render_800x800_4096_steps_ms"] = inf.get("ms_per_image")
    dr = cfg.get("dropin_autograd_ms_per_step")
    if isinstance(dr, dict):
        flat["reference_loop_on_dropin_ms_per_step"] = dr.get("fused_adam_fp16_planes")
    for wl, rep in (cfg.get("other_workloads") or {}).items():
        flat[f"{wl}_ms_per_step"] = rep.get("ms_per_step")
    for k, v in (cfg.get("sections_ms") or {}).items():
        flat["ms_" + k] = _num(v)
    flat["detail"] = "gpurun_out/bench_detail.json"
    top = [{"section": e["section"], "ms": e["ms_per_step"], "bound": e["bound"], "achieved": _num(e["achieved"]),
            "unit": e["unit"], "frac": _num(e["frac"]), "traffic_over_algorithmic": e.get("traffic_over_algorithmic")}
           for e in rf["top"]]
    roofline = {k: _num(rf.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    if roofline["achieved"] is not None and roofline["peak"]:
        roofline["frac"] = roofline["achieved"] / roofline["peak"]          # exactly A / P of the printed A and P
    roofline.update({"kernel": " + ".join(SECTION_KERNELS[top_section(rf)]), "section": top_section(rf),
                     "launches_per_step": rf["launches_per_step"], "avg_launch_ms": _num(rf["avg_launch_ms"]),
                     "algorithmic_bytes_per_launch": _num(rf["algorithmic_bytes_per_launch"]),
                     "algorithmic_flops_per_launch": _num(rf["algorithmic_flops_per_launch"]),
                     "alone_frac": None if not rf.get("alone") else rf["alone"]["frac"],
                     "measured_copy_GB/s": rf.get("measured_copy_GB/s"), "top": top})
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = flat
    line["roofline"] = roofline
    cb = out.get("cpu_baseline")
    if cb is not None:
        line["cpu_baseline"] = {"value": _num(cb["value"]), "unit": cb["unit"], "cores": cb["cores"],
                                "host_threads": cb.get("host_threads"), "kind": cb["kind"],
                                "sample": cb["sample_short"]}
    return line


def top_section(rf):
    return rf["section"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="base", choices=sorted(WORKLOADS))
    ap.add_argument("--dist-mode", default="auto", choices=["auto", "sharded", "allreduce"],
                    help="auto: mode and exchange bands from the cost model (distributed.plan_exchange, TNL_XGMI_GBS)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="the steady-state steps of a density-grid period replayed as captured HIP graphs (TrainStep(graph=True)). "
                         "Off by default: measured at base, natural loop, steady positions 3.50 ms eager vs 3.61-3.68 ms "
                         "captured (the capture joins the side stream at the end of every step and pays an input copy, a fill "
                         "and a cross-stream wait per replay; the ~0.1 ms of dispatch gaps it removes do not cover that)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the measured configuration); gloo only to dry-run the N>1 code on one GPU")
    ap.add_argument("--same-device", action="store_true", help="all ranks on cuda:0 (dry-run with --backend gloo)")
    ap.add_argument("--sections", action="store_true", help="print a per-section time breakdown to stderr")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank its own 60 000 rays per step; strong: the 60 000 rays of a step split over the ranks")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary figures (whole-plane / fp32-plane step times, inference, live PMC traffic)")
    ap.add_argument("--single-rank-collectives", action="store_true",
                    help="one GPU, but the N-GPU code path: a process group of ONE rank over --backend and "
                         "TrainStep(single_rank_collectives=True) -- every collective of the sharded step is issued (RCCL "
                         "kernels on its own stream); what the exchange's launches cost without any wire")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # a rocprofv3 --pmc pass of pmc_traffic_live
    args = ap.parse_args()
    if args.pmc_child or args.single_rank_collectives:
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
    lone = world == 1 and args.single_rank_collectives
    if lone:
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        kw = {"device_id": device} if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, **kw)

    from trinerflet_amd import build as tbuild
    from trinerflet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        if rank == 0:
            tbuild.build()
        if world > 1:
            dist.barrier()

    global GRAPH
    GRAPH = world == 1 and args.graph
    model, ts, bitfield, N = build(args.workload, device, args.dist_mode if (world > 1 or lone) else None,
                                   **({"single_rank_collectives": True} if lone else {}))
    if os.environ.get("TNL_CLIP_FAR"):      # A/B: the in-order march of refresh steps clipped to the occupied box (default on)
        ts.clip_far_in_order = os.environ["TNL_CLIP_FAR"] != "0"
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
    # paths -- sample-budget update, partial ROI rebuild -- has happened once) and a second one, then park the long-lived Python objects
    # in the permanent generation so that a generation-2 collection (10-20 ms with the torch module tree) cannot land
    # inside the timed steps.  The W warm-up steps and the K timed steps follow unchanged; the timed window still
    # contains its grid refresh (every 16th step).  The set-up steps are instrumented (a HIP event at every section
    # boundary): they tell which section of the step is the longest, i.e. which one the timed steps put events around.
    for i in range(16):
        one_step(model, ts, bitfield, batches[i % nb], mean_count, batches[(i + 1) % nb])
    ts.section_events, ts.section_names = [], None
    for i in range(16):          # a second, steady-state period (its refresh step included, as in the timed steps)
        one_step(model, ts, bitfield, batches[i % nb], mean_count, batches[(i + 1) % nb])
    torch.cuda.synchronize()
    pre = ts.section_times("median")     # the median over the period's steps: one step that allocates does not pick the section
    dominant = max(SECTION_PREV, key=lambda k: pre.get(k, 0.0))
    if os.environ.get("TNL_BENCH_SECTION") in SECTION_PREV:          # experiments: force the section timed live
        dominant = os.environ["TNL_BENCH_SECTION"]
    ts.section_events = None
    gc.collect()
    gc.freeze()
    for i in range(args.warmup):
        one_step(model, ts, bitfield, batches[i % nb], mean_count, batches[(i + 1) % nb])

    # HIP events on the launch stream inside the timed steps: only the two boundaries around the dominant section's
    # launches (every recorded boundary costs the stream 6-8 us; all of them together were 1.3 % of a step)
    ts.section_events = []
    ts.section_names = {SECTION_PREV[dominant], dominant}
    if dominant == "idwt_fwd":
        ts.section_names.add("adam_catchup")          # a replay of the deferred pass may sit between "begin" and the rebuild
    if ts.graph:
        # captured steps: nothing can be timed INSIDE a replayed graph (an event recorded during the capture has no
        # elapsed time: hipErrorInvalidHandle), and an event between the stages would split the graph at a point where
        # the side stream is forked.  The timed region runs without events; the dominant section's duration comes from
        # the instrumented eager pass right after it (same process, same state, the same kernels with the same arguments).
        ts.section_events, ts.section_names = None, None
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
    captured_steps = None if not ts.graph else {
        "replays": ts.graph_replays, "captures": ts.graph_captures,
        "note": "TrainStep(graph=True): positions 1..14 of a density-grid period (not the refresh step, not the step whose "
                "optimiser pass fills the 16-slot ring and replays it) are replayed as captured HIP graphs, one per ring "
                "position; bit-identical to the eager launches (tests/test_graph_step_gpu.py)"}
    # the dominant section's time from the HIP events recorded inside the timed region ...
    dom_ms = ts.section_times().get(dominant, float("nan"))
    # ... and every section's, for the other figures, from an instrumented pass AFTER it (not part of the K steps)
    ts.section_events, ts.section_names = [], None
    for i in range(min(args.steps, 16)):
        j = args.warmup + args.steps + i
        one_step(model, ts, bitfield, batches[j % nb], mean_count, batches[(j + 1) % nb])
    torch.cuda.synchronize()
    sec = ts.section_times()
    sec_instrumented_dominant = sec.get(dominant)
    dom_timing = "HIP events on the launch stream inside the timed steps"
    if dom_ms == dom_ms:
        sec[dominant] = dom_ms
    else:       # captured steps: see above
        dom_ms = sec_instrumented_dominant
        dom_timing = ("HIP events on the launch stream in the instrumented eager pass of min(K, 16) steps right after the timed "
                      "region -- the timed region replays captured graphs, inside which nothing can be timed (an event "
                      "recorded during a capture has no elapsed time on HIP)")
    # the same sections with NOTHING beside them (the next batch's march + tile sort in order on the launch stream
    # instead of on the side stream): what each kernel does alone, for `roofline.alone` and `kernels[*].alone_ms`
    ts.section_events, ts.section_names = [], None
    ts._drop_prefetch()
    ts.overlap_march = False
    for i in range(min(args.steps, 8)):
        j = args.warmup + args.steps + 16 + i
        one_step(model, ts, bitfield, batches[j % nb], mean_count, None)
    torch.cuda.synchronize()
    sec_alone = ts.section_times()
    ts.overlap_march = not os.environ.get("TNL_NO_OVERLAP")
    ts.section_events = None
    # the driver's K = 20 timed steps hold one grid refresh (1 / 20) where a long run holds one per 16 steps: the same
    # loop over 32 steps = two whole density-grid periods, refreshes and replays inside (after the timed region)
    whole_periods_ms = time_steps(model, ts, bitfield, batches, mean_count, 32, setup=0) * 1e3
    samples_per_step = float(np.mean(counts))
    acct = adam_accounting(ts, world)
    adam_bytes, n_launch, deferred_coefs, band_share = acct["adam_bytes"], acct["n_launch"], acct["deferred_coefs"], acct["band_share"]
    live, rects, S_own = acct["live"], acct["rects"], acct["S_own"]
    tables = ts.last_live_bands if (ts.last_live_bands is not None and live[0] is ts.last_live[0]) else [None] * ts.J
    adam_deferred = None
    if any(lv is not None for lv in live):
        cu_ms = sec.get("adam_catchup", float("nan"))
        all_coefs = float(S_own * 3 * sum(ts.coef.params[lvl].shape[-1] ** 2 for lvl in range(ts.J)))
        adam_deferred = {
            "live_rectangles": [None if lv is None else {"level_size": ts.coef.params[k].shape[-1], "origin_x": lv[0:3],
                                                         "origin_y": lv[3:6], "width": lv[6], "height": lv[7]}
                                for k, lv in enumerate(live)],
            "band_pieces_share_of_rectangle": band_share,
            "levels_updated_inside_the_adjoint": list(acct["fused_levels"]),
            "deferred_share_of_coefficients": round(deferred_coefs / all_coefs, 4),
            "steps_deferred_in_timed_region": timed_deferred, "flushes_in_timed_region": timed_flushes,
            "catchup": None if cu_ms != cu_ms else {
                "ms": round(cu_ms, 4), "records": ts.last_flush_records, "bytes": 24.0 * deferred_coefs,
                "GB/s": round(24.0 * deferred_coefs / (cu_ms * 1e-3) / 1e9, 1)},
            "note": "coefficients outside a level's live rectangle (what the windowed plane rebuild reads + where the "
                    "windowed adjoint writes; where band_pieces_share_of_rectangle is given, only that share of it: per 8 "
                    "rows the columns the occupied cells' projection reaches, level by level) are neither read for a "
                    "sampled texel nor reached by a data gradient until the occupancy window changes; their Adam(+L1) steps are replayed in registers by k_adam_l1_catchup, all pending steps "
                    "in one 24-B/coefficient pass (bit-identical p, m, v).  Every replay the timed steps caused runs "
                    "inside the timed region (the ring holds 16 steps; a flush also precedes the clock's stop)."}

    spec, launches = section_specs(ts, model, samples_per_step, world, acct)
    Cc, Rr, Hh = ts.C, ts.R, ts.H
    kernels = {k: roof(spec, k, sec.get(k, float("nan"))) for k in spec}
    for k in kernels:
        a = roof(spec, k, sec_alone.get(k, float("nan")))
        if kernels[k] is not None and a is not None:
            kernels[k]["alone_ms"] = a["ms_per_step"]
            kernels[k]["alone_frac"] = round(a["frac"], 4)

    # ---- secondary figures, after the timed region (rank 0's GPU only; skipped by --no-extras and in multi-GPU runs)
    extras = {}
    # what the run used and what the cost model (trinerflet_amd.distributed.plan_exchange) predicts for
    # it: compare ms_predicted with ms_per_step of a multi-GPU run, ms_one_gpu with the one-GPU line
    dist_mode_used = ts.dist_mode
    exchange_plan = None
    if world > 1 or lone:
        from trinerflet_amd import distributed as D_
        tex = ts.R * ts.R if ts._roi is None else ts._roi[6] * ts._roi[7]
        plan = D_.plan_exchange(max(world, 1), 3 * ts.C, tex, samples_per_step, transports=(ts.grad_transport,))
        bands = ts._exchange_bands(ts._roi)
        exchange_plan = {"mode": dist_mode_used, "bands": 0 if bands is None else len(bands), "transport": ts.grad_transport,
                         "link_GBs_assumed": plan.get("link_gbs"), "ms_predicted": None if world == 1 else round(plan["ms"], 3),
                         "ms_one_gpu_model": round(plan["ms_one_gpu"], 3)}
    roi_window = None if ts._roi is None else {"origin_x": ts._roi[0:3], "origin_y": ts._roi[3:6], "width": ts._roi[6],
                                                "height": ts._roi[7], "of": ts.R,
                                                "note": "bounding window of the occupied cells per plane (r = 0.8 sphere): "
                                                        "the finest IDWT level, layout change, plane gradient and adjoint "
                                                        "touch only this window on the 15 of 16 steps without a grid refresh"}
    pmc, pmc_src = None, "not collected (--no-extras or multi-GPU run)"
    copy_gbs = None
    plc = ts.placement
    if world == 1 and not args.no_extras:
        copy_gbs = copy_bandwidth(device)
        extras["inference"] = inference_figure(model, device)
        pmc, pmc_src = pmc_traffic_live(args.workload)
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
            others[wl] = variant_report(wl, device, batches)
        extras["other_workloads"] = others
        # SURVEY.md 8(d)'s second occupancy: a thin shell r in [0.7, 0.8] (stresses the skipping; its own sample budget)
        shell = variant_report(args.workload, device, batches, shell=(0.8, 0.7))
        extras["thin_shell_ms_per_step"] = shell["ms_per_step"]
        extras["thin_shell"] = shell
        # what a main_nerf.py user gets WITHOUT swapping the loop for TrainStep: the reference's own loop
        # (utils.py:1134-1175) on the drop-in modules, through autograd (tools/bench_dropin.py)
        extras["dropin_autograd_ms_per_step"] = dropin_figure(args.workload, device, batches, mean_count)
        extras["variants_note"] = ("no_roi: every step rebuilds / differentiates whole planes (no occupancy window, no "
                                   "gradient-support rectangles); fp32_planes: the sampler reads fp32 planes as the "
                                   "reference's training does (SURVEY F9; implies whole planes); same rays, budget, "
                                   f"{k} steps without a grid refresh after a 17-step set-up")
        extras["trajectory"] = trajectory_figure(args.workload, device)

    def traffic_of(name):
        """Counter traffic of a section's kernels per step: (corrected bytes, raw bytes, per kernel) or None."""
        if pmc is None:
            return None
        f = w = 0.0
        per = {}
        for kn, v in pmc.items():
            if section_owns(name, kn):
                f += v.get("FETCH_SIZE", 0.0)
                w += v.get("WRITE_SIZE", 0.0)
                per[kn] = {"FETCH_SIZE_KB": round(v.get("FETCH_SIZE", 0.0), 1), "WRITE_SIZE_KB": round(v.get("WRITE_SIZE", 0.0), 1)}
        if not per:
            return None
        return {"bytes": (2.0 * f + w) * 1024.0, "bytes_uncorrected": (f + w) * 1024.0, "per_kernel": per}

    if rank == 0:
        C, R, scale, H, _, lam = WORKLOADS[args.workload]
        ms = elapsed / args.steps * 1e3
        survey_bytes = (44 + 2) * 3.0 * C * R * R + samples_per_step * (12 * C * 2 + 48 * C + 64) + 64.0 * N
        wire = None
        if world > 1:
            S_all = 3 * C
            if dist_mode_used == "sharded":
                win = (roi_window["width"] * roi_window["height"]) if roi_window else R * R
                rs = S_all * win * 4.0 * (world - 1) / world       # reduce-scatter of the fp32 plane-gradient window
                ag = S_all * win * 2.0 * (world - 1) / world       # all-gather of the rebuilt fp16 planes (window)
                wire = {"reduce_scatter_plane_grad_bytes": rs, "all_gather_planes_bytes": ag,
                        "note": "sent per rank and step without a grid refresh (whole planes on refresh steps); "
                                "plus 54 kB of MLP-gradient all-reduce and three scalars"}
            else:
                wire = {"all_reduce_plane_grad_bytes": 2.0 * S_all * R * R * 4.0 * (world - 1) / world}
        # the three longest sections, each with its binding roof and its counter traffic
        ranked = sorted((k for k in kernels if kernels[k] is not None), key=lambda k: -kernels[k]["ms_per_step"])
        top = []
        for k in ranked[:3]:
            e = dict(kernels[k])
            tr = traffic_of(k)
            e["traffic"] = None if tr is None else tr["bytes"]
            e["traffic_uncorrected"] = None if tr is None else tr["bytes_uncorrected"]
            e["traffic_per_kernel"] = None if tr is None else tr["per_kernel"]
            e["traffic_over_algorithmic"] = None if tr is None else round(tr["bytes"] / e["algorithmic_bytes"], 3)
            if tr is not None and args.workload != "tiny" and not 0.3 < e["traffic_over_algorithmic"] < 3:
                print(f"bench.py: section {k}: counter traffic / algorithmic bytes = {e['traffic_over_algorithmic']} "
                      f"(outside 0.3 .. 3: check SECTION_KERNELS against {sorted(tr['per_kernel'])})", file=sys.stderr)
            top.append(e)
        dom = dict(kernels[dominant]) if kernels.get(dominant) else {}
        dtr = traffic_of(dominant)
        n_dom = launches[dominant]
        step_traffic = None
        if pmc is not None:
            fsum = sum(v.get("FETCH_SIZE", 0.0) for v in pmc.values())
            wsum = sum(v.get("WRITE_SIZE", 0.0) for v in pmc.values())
            step_traffic = {"bytes": (2.0 * fsum + wsum) * 1024.0, "bytes_uncorrected": (fsum + wsum) * 1024.0,
                            "TB/s": round((2.0 * fsum + wsum) * 1024.0 / (ms * 1e-3) / 1e12, 3),
                            "TB/s_uncorrected": round((fsum + wsum) * 1024.0 / (ms * 1e-3) / 1e12, 3),
                            "note": "sum over every kernel of a steady-state step (main and side stream) of the rocprofv3 "
                                    "counters, over ms_per_step (which also carries 1/16 of a refresh step and the replay "
                                    "of the deferred pass, whose bytes are not in this sum); the factor 2 on FETCH_SIZE is "
                                    "calibrated for 16-B/lane streaming reads, the gather kernels lie between the two figures"}
        out = {
            "metric": "train rays/sec (whole node)", "value": N * world * args.steps / elapsed, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: 3x{C}ch x {R}^2 planes, bior6.8 scale {scale}, hidden {H}, "
                                   f"{N} rays/step/GPU, solid-sphere occupancy r=0.8 (re-imposed after each refresh), "
                                   f"fp16 planes + fp16 MFMA MLP, fp32 masters, Adam+L1; GT colours = an analytic function "
                                   f"of the ray direction (synthetic.target_colors), not a second seeded field's render",
                       "rays_per_step_per_gpu": N, "rays_per_step_global": n_global,
                       "samples_per_step_per_gpu": samples_per_step,
                       "sample_budget_M": mean_count, "parallelism": f"ray-dp{world}" + (f"+{dist_mode_used}" if (world > 1 or lone) else ""),
                       "exchange_plan": exchange_plan,
                       "collectives": None if (world == 1 and not lone) else {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                                               "bytes_on_the_wire": wire},
                       "samples_per_sec": samples_per_step * world * args.steps / elapsed,
                       "ms_per_step_over_whole_periods": round(whole_periods_ms, 4),
                       "whole_periods_note": "32 steps = two whole density-grid periods (2 refreshes + their replays of the "
                                             "deferred pass), measured after the timed region: what a long run averages, "
                                             "where the timed K steps hold K / 16 refreshes only on average",
                       # SURVEY.md 8(d)'s byte count of the REFERENCE's step (every coefficient rebuilt, differentiated and
                       # updated every step: (44 + e) P + M (12 C e + 48 C + 64) + 64 N) over this step's time
                       "survey_step": {"bytes": survey_bytes, "TB/s_equivalent": round(survey_bytes / (ms * 1e-3) / 1e12, 3),
                                       "of_8_TB/s": round(survey_bytes / (ms * 1e-3) / 8e12, 3),
                                       "note": "bytes the reference's formulation of one step moves per GPU (SURVEY.md "
                                               "8(d), e = 2) divided by the measured step time; the step itself moves "
                                               "fewer (occupancy window, live rectangles of the optimiser pass)"},
                       "step_actual_traffic": step_traffic,
                       "sections_ms": {k: round(v, 4) for k, v in sec.items()},
                       "sections_note": f"{dominant}: HIP events inside the timed steps; the other sections: an "
                                        "instrumented pass after them (an event at every boundary costs 6-8 us)",
                       "kernels": kernels,
                       "captured_steps": captured_steps,
                       "roi_window": roi_window,
                       "adam_placement": plc,
                       "adam_deferred": adam_deferred,
                       **extras},
            "roofline": {"bound": dom.get("bound"), "section": dominant, "kernel": f"{' + '.join(SECTION_KERNELS[dominant])} (section '{dominant}' of the "
                                                              f"step: its {n_dom} launch(es) per step)",
                         "why_this_kernel": "the section with the largest per-step time (median over the 16 steps of the instrumented set-up pass): "
                                            + ", ".join(f"{k} {pre.get(k, float('nan')):.3f} ms" for k in sorted(SECTION_PREV, key=lambda k: -pre.get(k, 0.0))),
                         "achieved": dom.get("achieved"), "peak": dom.get("peak"), "unit": dom.get("unit"), "frac": dom.get("frac"),
                         "traffic": None if dtr is None else dtr["bytes"] / n_dom,
                         "traffic_uncorrected": None if dtr is None else dtr["bytes_uncorrected"] / n_dom,
                         "traffic_source": pmc_src,
                         "algorithmic_bytes_per_launch": spec[dominant]["bytes"] / n_dom,
                         "algorithmic_flops_per_launch": spec[dominant]["flops"] / n_dom,
                         "avg_launch_ms": dom_ms / n_dom, "launches_per_step": n_dom, "timing": dom_timing,
                         "per_step": {"algorithmic_bytes": spec[dominant]["bytes"], "algorithmic_flops": spec[dominant]["flops"],
                                      "traffic_bytes": None if dtr is None else dtr["bytes"], "ms": dom_ms,
                                      "ms_instrumented_pass": sec_instrumented_dominant,
                                      "GB/s": dom.get("GB/s"), "frac_hbm": dom.get("frac_hbm"),
                                      "mfma_TFLOP/s": dom.get("mfma_TFLOP/s"), "frac_mfma": dom.get("frac_mfma")},
                         "per_unit": spec[dominant]["per_unit"],
                         "alone": None if not dom.get("alone_ms") else {
                             "ms_per_step": dom["alone_ms"], "frac": dom["alone_frac"],
                             "note": "the same section in steps whose side work (the next batch's march + tile sort) runs "
                                     "in order on the launch stream instead of beside it: the kernel by itself"},
                         "top": top,
                         "measured_copy_GB/s": None if copy_gbs is None else round(copy_gbs, 1),
                         "measured_copy_note": "tnl_copy_probe: 1 GiB float4 non-temporal streaming copy on this box, read + "
                                               "write bytes over HIP-event time (the guide's figure: 6290 GB/s); peaks: HBM3E "
                                               "8000 GB/s spec, dense fp16 MFMA 2500 TFLOP/s",
                         "note": "bound = the roof that takes longer at peak for the section's algorithmic bytes and flops "
                                 "(SURVEY.md 8(d) per-unit figures x the units one step processes); achieved = those over "
                                 "the section's duration measured with HIP events on the launch stream inside the timed "
                                 "steps (the section's launches back to back, per launch = / launches_per_step); "
                                 "field_bwd includes its k_slab_reduce launch (weight-gradient slabs)"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        if args.sections:
            print(json.dumps(sec, indent=1), file=sys.stderr)
        # The whole record (per-kernel tables, trajectory, inference, other workloads, notes) goes to a side file; the
        # stdout line is the contract's keys only, flat and short (round 4's 22 KB line was not parsed by the driver).
        detail_path = os.environ.get("TNL_BENCH_DETAIL", os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
        try:
            os.makedirs(os.path.dirname(detail_path), exist_ok=True)
            with open(detail_path, "w") as fh:
                json.dump(out, fh, indent=1)
            print(f"bench.py: full record -> {detail_path}", file=sys.stderr)
        except OSError as e:
            print(f"bench.py: could not write {detail_path}: {e}", file=sys.stderr)
        sys.stderr.flush()
        line = json.dumps(compact_line(out))
        assert len(line) < 4096, len(line)
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    if world > 1 or lone:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
