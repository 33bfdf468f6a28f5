"""Tile-sorted plane-gradient reduction alone (csrc/scatter.hip) at a README geometry, on random samples inside the bench's
occupancy sphere: whole planes (autograd path: zero fill + prezeroed reduce) against the compact window (TrainStep).

    python tools/bench_tile_reduce.py [--C 32] [--R 2048] [--M 4650000]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--C", type=int, default=32)
    ap.add_argument("--R", type=int, default=2048)
    ap.add_argument("--M", type=int, default=4650000)
    a = ap.parse_args()
    from trinerflet_amd import _lib as L
    from trinerflet_amd.nerf import field as F_
    dev = torch.device("cuda:0")
    C, R, M, bound = a.C, a.R, a.M, 1.5
    g = torch.Generator(device=dev).manual_seed(0)
    u = torch.randn(M, 3, device=dev, generator=g)
    u = u / u.norm(dim=1, keepdim=True) * 0.8 * torch.rand(M, 1, device=dev, generator=g) ** (1 / 3)
    xyz = u.contiguous()
    dfeat = (torch.randn(3, M, C, device=dev, generator=g) * 1e-2).to(torch.float16)
    lib = L.lib()
    ws = F_.plane_grad_sort_workspace(M, R, dev)
    sort = lambda: L.check(lib.tnl_plane_grad_sort(L.ptr(xyz), L.f32(bound), L.u32(M), L.ptr(None), L.u32(R), L.ptr(ws),
                                                   L.stream()), "sort")
    sort()
    lo = int((1 - 0.8 / bound) / 2 * (R - 1)) // 64 * 64
    hi = min((int((1 + 0.8 / bound) / 2 * (R - 1)) + 3 + 63) // 64 * 64, R)
    roi = [lo] * 6 + [hi - lo, hi - lo, C, 0]
    whole = torch.zeros(3, C, R, R, device=dev)
    comp = torch.zeros(3 * C, hi - lo, hi - lo, device=dev)

    def reduce(out, layout, r):
        L.check(lib.tnl_plane_grad_reduce(L.ptr(dfeat), L.ptr(xyz), L.f32(bound), L.u32(M), L.u32(C), L.u32(R), L.f32(1.0),
                                          L.ptr(out), L.i32(layout), L.ptr(None), L.roi_array(r), L.ptr(ws), L.stream()),
                "reduce")
    print(f"C {C} R {R} M {M} window {roi[6]}x{roi[7]} at {lo}")
    print(f"sort                       {timed(sort):.3f} ms")
    print(f"zero fill whole            {timed(lambda: whole.zero_()):.3f} ms")
    print(f"reduce whole cm            {timed(lambda: reduce(whole, 1, None)):.3f} ms")
    print(f"reduce whole cm prezeroed  {timed(lambda: reduce(whole, 3, None)):.3f} ms")
    tm = torch.zeros(3, R, R, C, device=dev)
    print(f"reduce whole tm prezeroed  {timed(lambda: reduce(tm, 2, None)):.3f} ms")
    del tm
    print(f"reduce window compact      {timed(lambda: reduce(comp, 1, roi)):.3f} ms")
    ref = torch.zeros_like(whole)
    reduce(ref, 3, None)
    reduce(comp, 1, roi)
    c4 = comp.view(3, C, hi - lo, hi - lo)
    print("window == whole inside:", bool(torch.equal(ref[:, :, lo:hi, lo:hi], c4)))


if __name__ == "__main__":
    main()
