import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
import trinerflet_amd._lib as L
from trinerflet_amd import raymarching, synthetic
import os
lib = L.lib(); dev = torch.device('cuda:0')
lib.tnl_march_count_form(L.i32(int(os.environ.get('TNL_COUNT_FORM', '0'))))   # 0 wavefront per ray, 1 one ray per lane
for (bound, Cc, Hg, N) in ((1.0, 1, 128, 61440), (2.0, 2, 128, 61440), (1.5, 2, 128, 60000)):   # the last: the base workload's
    max_steps = 1024
    rng = np.random.default_rng(11)
    poses = synthetic.hemisphere_poses(100, seed=2)
    pix = np.stack([rng.integers(0, 100, N), rng.integers(0, 800 * 800, N)], -1)
    o_np, d_np = synthetic.get_rays(poses, pix)
    bits_np = synthetic.sphere_bitfield(Hg, Cc, bound, 0.8, 0.0)
    o, d, bits = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (o_np, d_np, bits_np))
    aabb = torch.tensor([-bound] * 3 + [bound] * 3, device=dev)
    nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.2)
    noise = torch.from_numpy(rng.random(N).astype(np.float32)).to(dev)
    nws = lib.tnl_march_rays_train_workspace_rec(L.u32(N), L.u32(max_steps))
    ws = torch.empty(nws, dtype=torch.int32, device=dev)
    M = 6_000_000
    xyzs, dirs, deltas = torch.empty(M, 3, device=dev), torch.empty(M, 3, device=dev), torch.empty(M, 2, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    def run():
        counter.zero_()
        L.check(lib.tnl_march_rays_train(L.ptr(o), L.ptr(d), L.ptr(bits), L.f32(bound), L.f32(0.0), L.u32(max_steps), L.u32(N), L.u32(Cc), L.u32(Hg), L.u32(M), L.ptr(nears), L.ptr(fars), L.ptr(xyzs), L.ptr(dirs), L.ptr(deltas), L.ptr(rays), L.ptr(counter), L.ptr(noise), L.ptr(ws), L.u32(nws), L.stream()), "m")
    for masked in (0,):
        for _ in range(3): run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): run()
        b.record(); torch.cuda.synchronize()
        print(f"bound {bound} masked {masked}: {a.elapsed_time(b)/20*1e3:.1f} us per march, samples {int(counter[0])}")
