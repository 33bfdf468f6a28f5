"""Host time to ENQUEUE one TrainStep.step at the base size vs the device time of the step (is the loop launch-bound?).
GPU box: PYTHONPATH=. python tools/host_overhead.py"""
import time
import numpy as np
import torch
import bench

dev = torch.device("cuda:0")
model, ts, bitfield, N = bench.build("base", dev, None)
batches = bench.make_batches(4, N, 0, dev)
model.mean_count = 0
counts = []
for b in batches:
    bench.one_step(model, ts, bitfield, b, 0)
    counts.append(int(ts.last["counter"][0].item()))
mc = int(max(counts) * 1.02)
model.mean_count = mc
for i in range(16):
    bench.one_step(model, ts, bitfield, batches[i % 4], mc, batches[(i + 1) % 4])
torch.cuda.synchronize()
host = []
t_all = time.perf_counter()
for i in range(64):
    t0 = time.perf_counter()
    bench.one_step(model, ts, bitfield, batches[i % 4], mc, batches[(i + 1) % 4])
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all
host = np.array(host) * 1e3
print(f"host enqueue per step: median {np.median(host):.2f} ms, p90 {np.percentile(host, 90):.2f} ms; "
      f"wall per step {t_all / 64 * 1e3:.2f} ms")
