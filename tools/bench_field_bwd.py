"""Times the fused field's backward kernel alone (binned mode: fp16 dF out, weight-gradient slabs) on synthetic samples
of the base / small / large geometry: python tools/bench_field_bwd.py [workload] [M].  HIP events on the launch stream."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from trinerflet_amd.nerf import field as F_   # noqa: E402

GEOM = {"base": (32, 64, 2048), "small": (16, 64, 1024), "large": (48, 128, 2048)}


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "base"
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 4_650_000
    C, H, R = GEOM[wl]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    tm = (torch.randn(3, R, R, C, generator=g) * 0.3).to(torch.float16).to(dev)
    xyz = ((torch.rand(M, 3, generator=g) * 2 - 1) * 0.8).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1).to(dev)
    shapes = [(H, 3 * C), (16, H), (H, 31), (H, H), (3, H)]
    W = [((torch.rand(s, generator=g) * 2 - 1) / s[1] ** 0.5).to(dev) for s in shapes]
    packed = F_.pack_weights(*W, C, H)
    sigma, rgb, feats = F_.field_forward(tm, xyz, dirs, packed, 1.5, C, R, H, save_feats=True)
    gs = torch.randn(M, generator=g).to(dev) * 1e-3
    gc = torch.randn(M, 3, generator=g).to(dev) * 1e-3
    gradW = torch.zeros(sum(a * b for a, b in shapes), device=dev)
    dfeat = torch.empty(3, M, C, dtype=torch.float16, device=dev)
    g_cm = torch.empty(1, device=dev)

    def run():
        F_.field_backward(gs, gc, sigma, None, feats, xyz, dirs, packed, 1.5, C, R, H, g_cm, gradW, dfeat=dfeat)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            run()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) / 5)
    ms = min(ts)
    mac = 3 * C * H + 16 * H + 31 * H + H * H + 3 * H
    print(f"{wl} M={M} backward {ms:.4f} ms (min of 5x5; all {[round(t, 4) for t in ts]})  "
          f"{6.0 * mac * M / ms / 1e9:.1f} TFLOP/s  {M * (12 * C + 40) / ms / 1e6:.0f} GB/s  dW checksum {float(gradW.abs().sum()):.6g}")


if __name__ == "__main__":
    main()
