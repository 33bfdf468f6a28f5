"""Does the Adam pass's time depend on the DATA in the arrays (toggle-rate dependent clocks / compression), not on
where they live?  One quartet, contents replaced in place between timings.
usage (GPU box): PYTHONPATH=. python tools/adam_data.py"""
import torch

from trinerflet_amd import _lib as L

N = 402_653_184
dev = torch.device("cuda:0")
lib = L.lib()
steps = torch.ones(1, device=dev)
found = torch.zeros(1, device=dev)
skip = torch.ones(1, device=dev)


def adam(p, g, m, v, reps=5, lr=0.0, fi=found):
    def run():
        L.check(lib.tnl_adam_l1_step_dev(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.u64(p.numel()), L.f32(lr), L.ptr(steps),
                                         L.f32(0.9), L.f32(0.99), L.f32(1e-15), L.f32(1.0), None, L.f32(0.0),
                                         L.ptr(fi), None, L.i32(0), L.stream()), "adam")
    run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        run()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for trial in range(3):
    q = [torch.empty(N, device=dev) for _ in range(4)]
    res = []
    for name, fill in (("all zero", lambda t: t.zero_()), ("N(0,1e-3)", lambda t: t.normal_(0, 1e-3)),
                       ("N(0,1)", lambda t: t.normal_(0, 1.0)), ("uniform bits", lambda t: t.view(torch.int32).random_(-2 ** 31, 2 ** 31 - 1)),
                       ("N(0,1e-3) again", lambda t: t.normal_(0, 1e-3))):
        for t in q:
            fill(t)
        if name == "uniform bits":
            for t in q:
                t.nan_to_num_(0.0, 1.0, -1.0).clamp_(-1e6, 1e6)
        q[3].abs_()
        res.append(f"{name}: {adam(*q):.3f}")
    # only p large, the rest small -- and the reverse
    for t in q:
        t.normal_(0, 1e-3)
    q[3].abs_()
    q[0].normal_(0, 1.0)
    res.append(f"p~N(0,1), g,m,v small: {adam(*q):.3f}")
    q[0].normal_(0, 1e-3); q[2].normal_(0, 1.0)
    res.append(f"m~N(0,1), rest small: {adam(*q):.3f}")
    res.append(f"skip path (found_inf=1: reads p only): {adam(*q, fi=skip):.3f}")
    print(f"quartet {trial}: " + " | ".join(res))
    del q
