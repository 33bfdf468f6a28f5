"""800x800 render at the large geometry (C 48, hidden 128): loop vs one-kernel render.  PYTHONPATH=. python tools/bench_render_large.py"""
import time, numpy as np, torch
from trinerflet_amd import synthetic
from trinerflet_amd.nerf.network import NeRFNetwork
dev = torch.device("cuda:0")
m = NeRFNetwork(encoding="triplane_wavelet", bound=1.5, cuda_ray=True, density_thresh=10, hidden_dim=128, hidden_dim_color=128,
                triplane_channels=48, triplane_resolution=2048, triplane_wavelet_levels=32, wavelet_type="bior6.8").to(dev)
synthetic.init_field_parameters(m, seed=0)
m.density_bitfield.copy_(torch.from_numpy(synthetic.sphere_bitfield(128, m.cascade, 1.5, 0.8, 0.0)).to(dev))
m.eval()
poses = synthetic.hemisphere_poses(1, seed=3)
pix = np.stack([np.zeros(640000, np.int64), np.arange(640000)], -1)
o, d = synthetic.get_rays(poses, pix)
o, d = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
with torch.no_grad():
    for max_steps in (1024, 4096):
        for tag, kw in (("loop", dict(device_loop=True)), ("one kernel", dict())):
            for rep in range(4):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                m.render(o, d, staged=True, bg_color=0, perturb=False, max_steps=max_steps, **kw)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"large max_steps {max_steps} {tag}: {dt * 1e3:.1f} ms")
