"""The Adam pass as the step launches it (one launch per wavelet level, rectangle-aware) against one launch over the
whole flat buffers, on several freshly allocated quartets.  usage (GPU box): PYTHONPATH=. python tools/adam_levels.py"""
import ctypes
import sys

import torch

from trinerflet_amd import _lib as L

C, J, n0 = 32, 5, 64
S = 3 * C
dev = torch.device("cuda:0")
lib = L.lib()
sizes = [S * 3 * (n0 << l) ** 2 for l in range(J)]
offs = [sum(sizes[:l]) for l in range(J)]
N = sum(sizes)
steps = torch.ones(1, device=dev)
found = torch.zeros(1, device=dev)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def whole(q):
    p, g, m, v = q
    L.check(lib.tnl_adam_l1_step_dev(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.u64(N), L.f32(0.0), L.ptr(steps), L.f32(0.9),
                                     L.f32(0.99), L.f32(1e-15), L.f32(1.0), None, L.f32(0.0), L.ptr(found), None, L.i32(0),
                                     L.stream()), "adam")


def levels(q, rect_frac):
    p, g, m, v = q
    for l in range(J):
        n = n0 << l
        o = offs[l]
        if rect_frac is None:
            L.check(lib.tnl_adam_l1_step_dev(L.ptr(p[o:]), L.ptr(g[o:]), L.ptr(m[o:]), L.ptr(v[o:]), L.u64(sizes[l]), L.f32(0.0),
                                             L.ptr(steps), L.f32(0.9), L.f32(0.99), L.f32(1e-15), L.f32(1.0), None, L.f32(0.0),
                                             L.ptr(found), None, L.i32(0), L.stream()), "adam")
        else:
            w = max(32, int(n * rect_frac) // 32 * 32)
            o0 = (n - w) // 2 // 32 * 32
            rect = (ctypes.c_int32 * 8)(o0, o0, o0, o0, o0, o0, w, w)
            L.check(lib.tnl_adam_l1_step_rect(L.ptr(p[o:]), L.ptr(g[o:]), L.ptr(m[o:]), L.ptr(v[o:]), L.u32(S), L.u32(3), L.u32(n),
                                              L.u32(C), L.u32(0), rect, L.f32(0.0), L.ptr(steps), L.f32(0.9), L.f32(0.99),
                                              L.f32(1e-15), L.f32(1.0), None, L.f32(0.0), L.ptr(found), None, L.stream()), "rect")


sets = [[torch.zeros(N, device=dev) for _ in range(4)] for _ in range(K)]
for k, q in enumerate(sets):
    print(f"set {k}: whole {timeit(lambda: whole(q)):.3f} | per level {timeit(lambda: levels(q, None)):.3f} | "
          f"per level, rect 0.6 {timeit(lambda: levels(q, 0.6)):.3f} | finest level alone "
          f"{timeit(lambda: L.check(lib.tnl_adam_l1_step_dev(L.ptr(q[0][offs[4]:]), L.ptr(q[1][offs[4]:]), L.ptr(q[2][offs[4]:]), L.ptr(q[3][offs[4]:]), L.u64(sizes[4]), L.f32(0.0), L.ptr(steps), L.f32(0.9), L.f32(0.99), L.f32(1e-15), L.f32(1.0), None, L.f32(0.0), L.ptr(found), None, L.i32(0), L.stream()), 'a')):.3f}"
          f" (= {sizes[4] / N:.2f} of the elements)")
