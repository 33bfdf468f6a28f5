"""Step time by position inside a density-grid period (refresh at position 0, 16 steps per period).

    python tools/period_profile.py [--workload base] [--periods 6]

Uses bench.py's own model / batches / step loop.  One HIP event per step boundary on the launch stream; prints the mean and
the minimum over the periods for every position, the steady-state mean (positions 2..15) and what positions 0 and 1 cost
beyond it.  Position 0 = the refresh step (replay of the deferred optimiser steps, whole planes, grid update, march in
order); position 1 = the first step under the new occupancy window (unfused adjoint, new tables)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="base")
    ap.add_argument("--periods", type=int, default=6)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    bench.GRAPH = False
    model, ts, bitfield, N = bench.build(args.workload, dev, None)
    batches = bench.make_batches(4, N, 0, dev)
    model.mean_count = 0
    counts = []
    for b in batches:
        bench.one_step(model, ts, bitfield, b, 0)
        counts.append(int(ts.last["counter"][0].item()))
    mean_count = int(max(counts) * 1.02)
    model.mean_count = mean_count
    nb = len(batches)
    P = ts.update_extra_interval
    i = 0
    while ts.global_step % P != 0 or i < 2 * P:            # two warm periods, then aligned to a period start
        bench.one_step(model, ts, bitfield, batches[i % nb], mean_count, batches[(i + 1) % nb])
        i += 1
    torch.cuda.synchronize()
    n = args.periods * P
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for k in range(n):
        bench.one_step(model, ts, bitfield, batches[(i + k) % nb], mean_count, batches[(i + k + 1) % nb])
        ev[k + 1].record()
    torch.cuda.synchronize()
    ms = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(n)]).reshape(args.periods, P)
    steady = ms[:, 2:].mean()
    print(f"workload {args.workload}: {args.periods} periods of {P} steps; mean over all {ms.mean():.3f} ms per step")
    for p in range(P):
        print(f"  position {p:2d}: mean {ms[:, p].mean():7.3f}  min {ms[:, p].min():7.3f}  max {ms[:, p].max():7.3f}")
    print(f"steady state (positions 2..{P - 1}): {steady:.3f} ms")
    print(f"position 0 costs {ms[:, 0].mean() - steady:+.3f} ms, position 1 {ms[:, 1].mean() - steady:+.3f} ms beyond a steady step: "
          f"{(ms[:, 0].mean() + ms[:, 1].mean() - 2 * steady) / P:.3f} ms per step amortised")


if __name__ == "__main__":
    main()
