"""Placement regimes of the Adam pass: the same kernel on freshly allocated (p, g, m, v) quartets runs in a fast or a
slow regime that follows the ALLOCATION.  This script measures, in one process:
  1. K quartets of 1.6-GB arrays (torch.empty = one hipMalloc each): k_adam_l1 time per quartet;
  2. every array alone (an in-place scale: one read + one write stream);
  3. quartets re-assembled from the fastest / slowest single arrays and from mixed quartets;
  4. quartets carved from ONE large allocation at several spacings.
usage (GPU box): PYTHONPATH=. python tools/adam_regimes.py [K] [numel]"""
import sys

import torch

from trinerflet_amd import _lib as L

K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = int(sys.argv[2]) if len(sys.argv) > 2 else 402_653_184
dev = torch.device("cuda:0")
lib = L.lib()
steps = torch.ones(1, device=dev)
found = torch.zeros(1, device=dev)


def adam(p, g, m, v, reps=5):
    def run():
        L.check(lib.tnl_adam_l1_step_dev(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.u64(p.numel()), L.f32(1e-2), L.ptr(steps),
                                         L.f32(0.9), L.f32(0.99), L.f32(1e-15), L.f32(1.0), None, L.f32(1e-6),
                                         L.ptr(found), None, L.i32(0), L.stream()), "adam")
    run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        run()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def alone(x, reps=5):
    x.mul_(1.0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        x.mul_(1.0)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


sets = []
for k in range(K):
    q = [torch.empty(N, device=dev) for _ in range(4)]
    for t in q:
        t.normal_(0, 1e-3)
    q[3].abs_()
    sets.append(q)
print("quartet times (ms):", " ".join(f"{adam(*q):.3f}" for q in sets))
print("again             :", " ".join(f"{adam(*q):.3f}" for q in sets))
single = [[alone(t) for t in q] for q in sets]
for k, row in enumerate(single):
    print(f"set {k}: arrays alone (ms, in-place scale) " + " ".join(f"{x:.3f}" for x in row),
          " addr " + " ".join(hex(t.data_ptr() >> 21) for t in sets[k]))
flat = sorted((single[k][j], k, j) for k in range(K) for j in range(4))
fast4 = [sets[k][j] for _, k, j in flat[:4]]
slow4 = [sets[k][j] for _, k, j in flat[-4:]]
fast4[3].abs_(); slow4[3].abs_()
print(f"quartet of the 4 fastest single arrays: {adam(*fast4):.3f} ms; of the 4 slowest: {adam(*slow4):.3f} ms")
times = [adam(*q) for q in sets]
kf, ks = min(range(K), key=lambda k: times[k]), max(range(K), key=lambda k: times[k])
print(f"fastest set {kf} {times[kf]:.3f}, slowest set {ks} {times[ks]:.3f}")
for name, q in (("p,g fast + m,v slow", [sets[kf][0], sets[kf][1], sets[ks][2], sets[ks][3]]),
                ("p,g slow + m,v fast", [sets[ks][0], sets[ks][1], sets[kf][2], sets[kf][3]]),
                ("p fast, rest slow", [sets[kf][0], sets[ks][1], sets[ks][2], sets[ks][3]]),
                ("p slow, rest fast", [sets[ks][0], sets[kf][1], sets[kf][2], sets[kf][3]])):
    q[3].abs_()
    print(f"  {name}: {adam(*q):.3f} ms")
# is "slow" a property of single arrays?  every array in turn in one role, the other three roles from the fastest set
ROLES = [int(c) for c in sys.argv[3]] if len(sys.argv) > 3 else range(4)
for role in ROLES:
    row = []
    for k in range(K):
        for j in range(4):
            q = list(sets[kf])
            if sets[k][j] is q[0] or sets[k][j] is q[1] or sets[k][j] is q[2] or sets[k][j] is q[3]:
                row.append(float("nan"))
                continue
            q[role] = sets[k][j]
            if role == 3:
                q[3].abs_()
            row.append(adam(*q, reps=3))
    print(f"role {'pgmv'[role]}: " + " ".join(f"{x:.2f}" for x in row))
del sets, fast4, slow4
torch.cuda.empty_cache()
big = torch.empty(4 * N + 64 * 2 ** 20, device=dev)
big.normal_(0, 1e-3).abs_()
for pad in (0, 4096, 65536 + 4096, 2 ** 20 + 8192, 3 * 2 ** 20 + 64):
    q = [big[j * (N + pad):j * (N + pad) + N] for j in range(4)]
    print(f"one allocation, spacing +{pad * 4} B: {adam(*q):.3f} ms")
