"""The `--test` render (800 x 800, max_steps 4096; main_nerf.py:190-194, renderer.py:324-374) on a TRAINED field: the base
trajectory's 512 steps (tools/trajectory.py), then the one-kernel render timed for several budgets of marching work per
ray and trip (tnl_render_work; 0 = unbounded, the form of rounds 4-5), the images compared bit for bit.
usage (GPU box): PYTHONPATH=. python tools/bench_render_trained.py [workload] [steps]"""
import importlib.util
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trinerflet_amd import _lib as L, synthetic  # noqa: E402

spec = importlib.util.spec_from_file_location("tnl_trajectory", os.path.join(ROOT, "tools", "trajectory.py"))
T = importlib.util.module_from_spec(spec)
spec.loader.exec_module(T)
wl = sys.argv[1] if len(sys.argv) > 1 else "base"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
rep = T.run_fused(wl, dev, steps, 60000)
model = rep.pop("_model")
model.eval()
poses = synthetic.hemisphere_poses(3, seed=3)
out = {"workload": wl, "train_steps": steps, "held_out_psnr_db": rep.get("held_out_psnr_db"), "budgets": {}}
ref = None
for work in (0, 32, 64, 96, 128, 192, 256, 512):
    L.lib().tnl_render_work(L.i32(work))
    ms = []
    imgs = []
    for k in range(3):
        pix = np.stack([np.full(640000, k, np.int64), np.arange(640000)], -1)
        o, d = synthetic.get_rays(poses, pix)
        o, d = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
        with torch.no_grad():
            model.render(o, d, staged=True, bg_color=0, perturb=False, max_steps=4096)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = model.render(o, d, staged=True, bg_color=0, perturb=False, max_steps=4096)
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
        imgs.append((res["image"].clone(), res["weights_sum"].clone()))
    if ref is None:
        ref = imgs
    same = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(ref, imgs))
    out["budgets"][str(work)] = {"ms_per_image": round(float(np.mean(ms)), 3), "bit_identical_to_unbounded": same}
    print(work, out["budgets"][str(work)], file=sys.stderr, flush=True)
L.lib().tnl_render_work(L.i32(96))
print(json.dumps(out))
