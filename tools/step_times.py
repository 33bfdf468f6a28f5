"""GPU box: PYTHONPATH=. python tools/step_times.py [clip_far 0|1] -- per-step milliseconds of the base workload from HIP events
at the step boundaries (64 steps after the bench's set-up), to see which steps carry what."""
import sys
import numpy as np
import torch
import bench

clip = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
model, ts, bitfield, N = bench.build("base", dev, None)
ts.clip_far = bool(clip)
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
evs = [torch.cuda.Event(enable_timing=True) for _ in range(65)]
evs[0].record()
for i in range(64):
    bench.one_step(model, ts, bitfield, batches[i % 4], mc, batches[(i + 1) % 4])
    evs[i + 1].record()
torch.cuda.synchronize()
t = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(64)])
print("clip_far", clip, "mean", round(float(t.mean()), 3), "median", round(float(np.median(t)), 3))
print(np.round(t, 2).tolist())
