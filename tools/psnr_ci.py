"""PSNR evidence with a confidence interval (VERDICT r04 "next" 8; measure: reconstruction/nerf/utils.py:245-285).

Per scene (sphere, detail) and per seed (batch order, perturbation noise, initialisation): the fused fp16-plane TrainStep,
the reference's loop on the drop-in modules with fp32 planes + torch.optim.Adam (tools/trajectory.py::run_reference_loop),
and that loop with FusedAdamL1 + the windowed rebuild -- all with the PRODUCT-DEFAULT (unordered, atomics-ordered) tile
lists, base geometry, 512 steps from an untrained grid.  Every seed is reported (none left out).  For the paired
differences d_s = arm_s - reference_s: mean, standard deviation, and the 95 % confidence interval of the mean
(Student t, n - 1 degrees of freedom).  Written to profiles/r05_psnr_ci_<scene>.json; tests/test_trajectory_gpu.py checks
|mean| + half-width < 0.1 dB on the recorded file.

    PYTHONPATH=. python tools/psnr_ci.py --scene sphere --seeds 0 1 2 3 4 --out profiles/r05_psnr_ci_sphere.json
"""
import argparse
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# two-sided 95 % Student t quantiles by degrees of freedom
T975 = {1: 12.706, 2: 4.303, 3: 3.182, 4: 2.776, 5: 2.571, 6: 2.447, 7: 2.365, 8: 2.306, 9: 2.262, 10: 2.228, 11: 2.201,
        12: 2.179, 13: 2.160, 14: 2.145, 15: 2.131, 16: 2.120, 17: 2.110, 18: 2.101, 19: 2.093, 20: 2.086, 21: 2.080, 22: 2.074,
        23: 2.069, 24: 2.064, 25: 2.060, 26: 2.056, 27: 2.052, 28: 2.048, 29: 2.045, 30: 2.042, 31: 2.040}


def ci(diffs):
    n = len(diffs)
    mean = sum(diffs) / n
    sd = math.sqrt(sum((d - mean) ** 2 for d in diffs) / (n - 1)) if n > 1 else float("nan")
    half = T975.get(n - 1, 1.96) * sd / math.sqrt(n) if n > 1 else float("nan")
    return {"n": n, "mean_db": round(mean, 4), "sd_db": round(sd, 4), "ci95_half_width_db": round(half, 4),
            "ci95_db": [round(mean - half, 4), round(mean + half, 4)], "abs_mean_plus_half_width_db": round(abs(mean) + half, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="sphere", choices=["sphere", "detail"])
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1, 2, 3, 4])
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--workload", default="base")
    ap.add_argument("--out", default=None)
    ap.add_argument("--arm", default=None, choices=[None, "bf16_transport"],
                    help="bf16_transport: the fused step against ITSELF with the plane gradient rounded to bfloat16 before "
                         "the adjoint (TrainStep(grad_transport='bf16') in one process = what a slice's owner receives from "
                         "one rank: the gate of the multi-GPU bf16 exchange, SURVEY.md 8(e))")
    args = ap.parse_args()
    import importlib.util
    spec = importlib.util.spec_from_file_location("tnl_trajectory", os.path.join(ROOT, "tools", "trajectory.py"))
    T = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(T)
    dev = torch.device("cuda:0")
    scene = T.make_scene(dev, scene=args.scene)
    rows = []
    if args.arm == "bf16_transport":
        for seed in args.seeds:
            batches = T.batches_of(scene[0], args.steps, 60000, seed)
            a32 = T.run_fused(args.workload, dev, args.steps, 60000, scene, batches, seed=seed)
            a32.pop("_model")
            torch.cuda.empty_cache()
            a16 = T.run_fused(args.workload, dev, args.steps, 60000, scene, batches, seed=seed,
                              ts_kwargs={"grad_transport": "bf16"})
            a16.pop("_model")
            torch.cuda.empty_cache()
            rows.append({"seed": seed, "fp32_transport_db": a32["held_out_psnr_db"], "bf16_transport_db": a16["held_out_psnr_db"]})
            print(rows[-1], file=sys.stderr, flush=True)
        rep = {"scene": args.scene, "workload": args.workload, "steps": args.steps, "reductions": "unordered (product default)",
               "runs": rows, "bf16_minus_fp32_transport": ci([r["bf16_transport_db"] - r["fp32_transport_db"] for r in rows]),
               "note": "paired by seed; both arms the fused TrainStep, one process; bf16 arm: the plane-gradient window rounded "
                       "to bfloat16 once before the adjoint (the rounding every rank's contribution gets under "
                       "grad_transport='bf16'; the owner's fp32 accumulation adds none)"}
        s_ = json.dumps(rep, indent=1)
        print(s_)
        if args.out:
            with open(args.out, "w") as f:
                f.write(s_)
        return
    for seed in args.seeds:
        batches = T.batches_of(scene[0], args.steps, 60000, seed)
        fused = T.run_fused(args.workload, dev, args.steps, 60000, scene, batches, seed=seed)
        fused.pop("_model")
        torch.cuda.empty_cache()
        ref = T.run_reference_loop(args.workload, dev, args.steps, 60000, scene, batches, seed=seed)
        ref.pop("_model")
        torch.cuda.empty_cache()
        fast = T.run_reference_loop(args.workload, dev, args.steps, 60000, scene, batches, seed=seed, fast=True)
        fast.pop("_model")
        torch.cuda.empty_cache()
        rows.append({"seed": seed, "fused_db": fused["held_out_psnr_db"], "reference_loop_db": ref["held_out_psnr_db"],
                     "dropin_fast_loop_db": fast["held_out_psnr_db"], "fused_ms_per_step": fused["wall_ms_per_step"],
                     "reference_loop_ms_per_step": ref["wall_ms_per_step"], "dropin_fast_loop_ms_per_step": fast["wall_ms_per_step"]})
        print(rows[-1], file=sys.stderr, flush=True)
    rep = {"scene": args.scene, "workload": args.workload, "steps": args.steps, "reductions": "unordered (product default)",
           "runs": rows,
           "fused_minus_reference": ci([r["fused_db"] - r["reference_loop_db"] for r in rows]),
           "dropin_fast_minus_reference": ci([r["dropin_fast_loop_db"] - r["reference_loop_db"] for r in rows]),
           "note": "paired by seed (same batches, perturbation noise, initialisation, refresh draws); reference arm = the "
                   "reference Trainer's loop (utils.py:1134-1175) on the drop-in modules, fp32 planes, torch.optim.Adam, "
                   "torch GradScaler; held-out PSNR = mean of per-image PSNRs over 4 unseen 400x400 cameras"}
    s = json.dumps(rep, indent=1)
    print(s)
    if args.out:
        with open(args.out, "w") as f:
            f.write(s + "\n")


if __name__ == "__main__":
    main()
