"""Host-side cost of enqueuing one TrainStep.step() (no synchronisation inside the loop) beside the GPU time per step:
PYTHONPATH=. python tools/host_step_time.py [workload]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "base"
dev = torch.device("cuda:0")
model, ts, bitfield, N = bench.build(workload, dev, None)
batches = bench.make_batches(4, N, 0, dev)
model.mean_count = 0
counts = []
for b in batches:
    bench.one_step(model, ts, bitfield, b, 0)
    counts.append(int(ts.last["counter"][0].item()))
mc = int(max(counts) * 1.02)
model.mean_count = mc
for i in range(40):
    bench.one_step(model, ts, bitfield, batches[i % 4], mc, batches[(i + 1) % 4])
torch.cuda.synchronize()
K = 64
host = []
t0 = time.perf_counter()
for i in range(K):
    a = time.perf_counter()
    bench.one_step(model, ts, bitfield, batches[i % 4], mc, batches[(i + 1) % 4])
    host.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
host.sort()
print(f"{workload}: enqueue loop {1e3 * (t1 - t0) / K:.3f} ms/step (median call {1e3 * host[K // 2]:.3f}, min {1e3 * host[0]:.3f}, "
      f"max {1e3 * host[-1]:.3f}); with the final sync {1e3 * (t2 - t0) / K:.3f} ms/step")
