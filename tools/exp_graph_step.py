"""Experiment: one steady-state TrainStep.step() captured as a HIP graph (torch.cuda.graph: stream capture of everything the
step launches on the current stream and on the side stream it forks and joins) and replayed, against the same step launched
eagerly.  The captured step's host-side arguments (learning rate, ring slot, batch) are frozen, so the replays are not a
valid training -- only their duration is of interest: what removing the per-launch dispatch of ~40 kernels would buy.

    python tools/exp_graph_step.py [--workload base] [--reps 48]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="base")
    ap.add_argument("--reps", type=int, default=48)
    a = ap.parse_args()
    import bench as B
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    model, ts, bitfield, N = B.build(a.workload, dev, None)
    ts.update_extra_interval = 0            # no density-grid refresh (host read-backs) inside the captured region
    batches = B.make_batches(2, N, 0, dev)
    model.mean_count = 0
    counts = []
    for i in range(4):
        B.one_step(model, ts, bitfield, batches[i % 2], 0)
        counts.append(int(ts.last["counter"][0]))
    mean_count = int(max(counts) * 1.02)
    model.mean_count = mean_count
    o, d, gt, nz = batches[0]

    def step():
        return ts.step(o, d, gt, noises=nz)      # no next_rays: the side work of THIS batch is forked and joined inside

    for _ in range(20):
        step()
    torch.cuda.synchronize()

    def timed(fn, reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    eager = timed(step, a.reps)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()                                   # warm-up on the capture stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            step()
    torch.cuda.synchronize()
    replay = timed(g.replay, a.reps)
    eager2 = timed(step, a.reps)
    print({"workload": a.workload, "eager_ms_per_step": round(eager, 3), "graph_replay_ms_per_step": round(replay, 3),
           "eager_again_ms_per_step": round(eager2, 3),
           "note": "the step without next_rays: this batch's march + tile sort on the side stream beside the plane rebuild"})
    # the bench's form: the NEXT batch's march + sort forked after the field backward, joined at the end of the captured
    # region (a graph cannot leave a forked stream open); the replays read the prefetch the last eager step left
    o2, d2, gt2, nz2 = batches[1]

    def step_pf(join):
        ts.step(o, d, gt, noises=nz, next_rays=(o, d, nz))
        if join:
            torch.cuda.current_stream().wait_stream(ts._side)
    for _ in range(8):
        step_pf(False)
    eager_pf = timed(lambda: step_pf(False), a.reps)
    eager_pf_join = timed(lambda: step_pf(True), a.reps)
    g2 = torch.cuda.CUDAGraph()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step_pf(True)
        torch.cuda.synchronize()
        keep = ts._prefetched
        with torch.cuda.graph(g2, stream=s):
            # the prefetch the last eager step left is complete (synchronised above); its events belong to uncaptured work
            # and may not be waited for inside the capture: stand-ins recorded inside it
            e = torch.cuda.Event()
            e.record()
            ts._prefetched = (keep[0], (keep[1][0], (e, e)), keep[2])
            step_pf(True)
        ts._prefetched = keep
    torch.cuda.synchronize()
    replay_pf = timed(g2.replay, a.reps)
    print({"prefetch_form": {"eager_ms_per_step": round(eager_pf, 3), "eager_joined_at_the_end": round(eager_pf_join, 3),
                             "graph_replay_ms_per_step": round(replay_pf, 3)}})


if __name__ == "__main__":
    main()
