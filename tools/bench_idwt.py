"""Micro-benchmark of the finest IDWT level (forward fp16 / adjoint) with and without an occupancy window.
usage (GPU box): python tools/bench_idwt.py [C] [n]"""
import sys
import torch
from trinerflet_amd import _lib as L
from trinerflet_amd.triplaneencoder import triplane_encoder as te

C = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
if len(sys.argv) > 3:   # walk_min_n: 1048576 forces the LDS-tiled kernels, 8 the column-walk kernels
    L.lib().tnl_idwt_set_walk_min_n(L.u32(int(sys.argv[3])))
for kv in sys.argv[4:]:  # key=value pairs of tnl_idwt_set_tuning (1: forward rows per phase, 2: XCD order, 3: rows per workgroup)
    k, v = kv.split("=")
    assert L.lib().tnl_idwt_set_tuning(int(k), int(v)) == 0
R, S, wid = 2 * n, 3 * C, 4
dev = torch.device("cuda:0")
lib = L.lib()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


x = torch.randn(3, C, n, n, device=dev)
yh = torch.randn(3, C, 3, n, n, device=dev)
dx = torch.empty(S, n, n, device=dev)
dyh = torch.empty(S, 3, n, n, device=dev)
for name, (rw, rh) in {"full": (0, 0), "56%x56%": (R * 9 // 16 // 64 * 64,) * 2, "64x64": (64, 64)}.items():
    if rw == 0:
        roi, g = None, torch.randn(S, R, R, device=dev)
    else:
        o = (R - rw) // 2 // 64 * 64
        roi = [o, o, o, o, o, o, rw, rh, C, 0]
        g = torch.randn(S, rh, rw, device=dev)
    ra = L.roi_array(roi)
    t_b = timeit(lambda: L.check(lib.tnl_idwt_level_backward_roi(L.ptr(g), L.u32(S), L.u32(n), L.i32(wid), L.ptr(dx),
                                                                 L.ptr(dyh), ra, L.stream()), "bwd"))
    if roi is None:
        t_f = timeit(lambda: te.idwt_level_half(x, yh, wid))
    else:
        t_f = timeit(lambda: te.idwt_level_half_roi(x, yh, wid, roi))
    print(f"{name:10s} adjoint {t_b:8.1f} us   forward(fp16) {t_f:8.1f} us")
z = torch.empty(S * 4 * n * n, device=dev)
print(f"torch zero-fill of the adjoint's output size ({z.numel() * 4 / 1e9:.2f} GB): {timeit(lambda: z.zero_()):8.1f} us")
print(f"torch copy of the same size: {timeit(lambda: z.copy_(dyh.view(-1)[:1].expand(1)) if False else dx.copy_(dx)):8.1f} us")
