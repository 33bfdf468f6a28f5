"""GPU box: python tools/micro/copy_shapes.py -- GB/s (read + write) of a streaming copy with 4 / 8 / 16 bytes per lane."""
import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "copy_shapes.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "copy_shapes.hip")])
lib = ctypes.CDLL(so)
n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for width in (4, 8, 16):
    for rows in (1, 8, 64):
        f = lambda: lib.copy_shape(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_uint64(n), width, rows, st)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{width:2d} B/lane, {rows:3d} pieces per workgroup: {2 * n / ms / 1e6:7.0f} GB/s")
