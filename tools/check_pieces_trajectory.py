"""Is a real trajectory (untrained grid, real refreshes, two cascades) the same training with and without the
occupancy pieces?  Runs the fused loop twice with the ordered plane-gradient reduction (deterministic=True) and reports
the first step whose rendered colours differ.  GPU box: PYTHONPATH=. python tools/check_pieces_trajectory.py [workload] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import trajectory as T  # noqa: E402
from trinerflet_amd.train import TrainStep  # noqa: E402


def run(workload, dev, steps, scene, batches, bands):
    train, valid = scene
    model, lam = T.make_model(workload, dev, seed=0)
    ts = TrainStep(model, lr=1e-2, wavelet_regularization=lam, iters=512, warmup_steps=0, fp16=True, background_color=0.0,
                   deterministic=True, live_bands=bands)
    model.mark_untrained_grid(train.poses, train.intrinsics)
    ts.invalidate_roi()
    torch.manual_seed(1234)
    sums, wins = [], []
    banded = 0
    for k, (o, d, gt, nz) in enumerate(batches[:steps]):
        nxt = batches[k + 1] if k + 1 < steps else None
        ts.step(o, d, gt, noises=nz, next_rays=None if nxt is None else (nxt[0], nxt[1], nxt[3]))
        sums.append((ts.last["image"].double().sum().item(), int(ts.last["counter"][0]), float(ts.last["mse"])))
        wins.append(None if ts._roi is None else tuple(ts._roi))
        banded += int(ts._pending > 0 and ts._live_bands is not None and any(t is not None for t in ts._live_bands))
    ts.flush_deferred()
    params = [p.detach().clone() for p in model.parameters()]
    bits = model.density_bitfield.clone()
    return sums, wins, params, bits, banded


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "small"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 160
    dev = torch.device("cuda:0")
    scene = T.make_scene(dev)
    batches = T.batches_of(scene[0], max(steps, 2), 60000)
    same_variant = len(sys.argv) > 3 and sys.argv[3] == "same"      # control: the SAME variant twice
    a = run(workload, dev, steps, scene, batches, False)
    torch.cuda.empty_cache()
    b = run(workload, dev, steps, scene, batches, not same_variant)
    first = next((k for k in range(steps) if a[0][k][:2] != b[0][k][:2]), None)      # colours and sample count (the MSE is a float-atomic sum)
    print("first differing step:", first, "| steps whose optimiser pass ran over band pieces:", b[4], "| windowed steps:",
          sum(w is not None for w in b[1]))
    if first is not None:
        for k in range(max(first - 2, 0), min(first + 3, steps)):
            print(k, "no pieces", a[0][k], "window", a[1][k], "| pieces", b[0][k], "window", b[1][k])
    same = all(torch.equal(x, y) for x, y in zip(a[2], b[2]))
    print("parameters after", steps, "steps bit-identical:", same, "| bitfields equal:", torch.equal(a[3], b[3]))
    if not same:
        for i, (x, y) in enumerate(zip(a[2], b[2])):
            if not torch.equal(x, y):
                print("  parameter", i, tuple(x.shape), "differs in", int((x != y).sum()), "elements, max |d|", float((x - y).abs().max()))


if __name__ == "__main__":
    main()
