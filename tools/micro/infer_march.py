"""GPU box: PYTHONPATH=. python tools/micro/infer_march.py -- where one iteration of the inference march spends its time:
640 000 alive rays, n_step = 1 / 8, rays inside the occupied ball; (a) as is, (b) far = 0 (no probe at all)."""
import numpy as np, torch
from trinerflet_amd import _lib as L, synthetic, raymarching
lib = L.lib(); dev = torch.device("cuda:0")
N, bound, Cc, Hg, max_steps = 640000, 1.5, 2, 128, 4096
poses = synthetic.hemisphere_poses(1, seed=3)
pix = np.stack([np.zeros(N, np.int64), np.arange(N)], -1)
o, d = (torch.from_numpy(a).to(dev) for a in synthetic.get_rays(poses, pix))
bits = torch.from_numpy(synthetic.sphere_bitfield(Hg, Cc, bound, 0.8, 0.0)).to(dev)
aabb = torch.tensor([-bound] * 3 + [bound] * 3, device=dev)
nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.2)
alive = torch.arange(N, dtype=torch.int32, device=dev)
cap = 8 * N + 128
xyzs, dirs = torch.empty(cap, 3, device=dev), torch.empty(cap, 3, device=dev)
deltas, tscr = torch.empty(cap, 2, device=dev), torch.empty(cap, device=dev)
def run(n_step, rays_t, far, scratch):
    state = torch.tensor([N, n_step, 0, N * n_step], dtype=torch.int32, device=dev)
    f = lambda: L.check(lib.tnl_march_rays_dev(L.ptr(state), L.u32(N), L.ptr(alive), L.ptr(rays_t), L.ptr(o), L.ptr(d), L.f32(bound),
        L.f32(0.0), L.u32(max_steps), L.u32(Cc), L.u32(Hg), L.ptr(bits), L.ptr(far), L.ptr(xyzs), L.ptr(dirs), L.ptr(deltas),
        None, L.ptr(scratch), L.u32(cap), L.stream()), "march")
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 20 * 1e3
mid = ((nears + fars) / 2).contiguous()          # a t inside the ball for the central rays
pm = o + mid[:, None] * d
inside = (pm.norm(dim=1) < 0.7).nonzero().squeeze(1).to(torch.int32)   # rays whose midpoint lies inside the occupied ball
print("rays with their midpoint inside the ball:", inside.numel())
N_all, alive_all = N, alive
for n_step in (1, 8):
    for name, rt, fr in (("from near (skips empty space first)", nears, fars), ("from the middle of the ray", mid, fars),
                         ("far = 0: no probe", mid, torch.zeros_like(fars)), ("middle, only rays inside the ball", mid, fars)):
        if name.startswith("middle, only"):
            alive, N = inside.contiguous(), inside.numel()
        else:
            alive, N = alive_all, N_all
        for sc_name, sc in (("record + emit", tscr), ("one kernel", None)):
            print(f"n_step {n_step}  {name:38s} {sc_name:14s}: {run(n_step, rt, fr, sc):7.1f} us")
