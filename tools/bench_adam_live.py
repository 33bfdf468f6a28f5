"""Times the step's coefficient pass alone (k_adam_record + k_adam_l1_live + the LL launch) at a bench workload, with the
band pieces and with whole live rectangles: PYTHONPATH=. python tools/bench_adam_live.py [workload]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "base"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    res = {}
    for bands in (True, False, True, False):
        model, ts, bitfield, N = bench.build(workload, dev, None)
        ts.live_bands = bands
        batches = bench.make_batches(2, N, 0, dev)
        model.mean_count = 0
        for i in range(3):
            bench.one_step(model, ts, bitfield, batches[i % 2], 0)
        torch.cuda.synchronize()
        assert ts._live is not None and ts._pending > 0
        found = torch.zeros(1, device=dev)
        inv = torch.ones(1, device=dev)
        rects = ts._rects
        s0, s1 = 0, 3 * ts.C

        def run():
            ts._pending = 1
            ts._adam_levels_live(1e-3, ts._defer_ctx[2], found, inv, s0, s1, rects)
        for _ in range(3):
            run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run()
        b.record()
        b.synchronize()
        ms = a.elapsed_time(b) / 20
        share = None if not bands else [None if t is None else round(t[1] * 4 / (lv[6] * lv[7]), 3)
                                        for t, lv in zip(ts._live_bands, ts._live)]
        print(f"bands={bands}: {ms:.4f} ms  pieces/rectangle {share}")
        res.setdefault(bands, []).append(ms)
        ts._pending = 0
        ts._live = None
        del model, ts
        torch.cuda.empty_cache()


if __name__ == "__main__" and not (len(sys.argv) > 2 and sys.argv[2] == "coarse"):
    main()


def coarse_beside():
    """Experiment: the coefficient pass with the three coarse forward levels of the NEXT rebuild running beside it on
    a second stream, against the two one after the other."""
    dev = torch.device("cuda:0")
    model, ts, bitfield, N = bench.build("base", dev, None)
    batches = bench.make_batches(2, N, 0, dev)
    model.mean_count = 0
    for i in range(3):
        bench.one_step(model, ts, bitfield, batches[i % 2], 0)
    torch.cuda.synchronize()
    found, inv = torch.zeros(1, device=dev), torch.ones(1, device=dev)
    rects, s0, s1 = ts._rects, 0, 3 * ts.C
    enc = ts.enc
    wins = ts._forward_windows()
    from trinerflet_amd.triplaneencoder.triplane_encoder import _IDWTLevel

    def coarse():
        x = enc.planes_features
        with torch.no_grad():
            for lvl in range(ts.J - 2):
                yh = enc.planes_features_wavelet_coefs[lvl]
                x = ts._idwt_level_win(x, yh, wins[lvl]) if wins[lvl] is not None else _IDWTLevel.apply(x, yh, enc.wave_id)
        return x

    def adam():
        ts._pending = 1
        ts._adam_levels_live(1e-3, ts._defer_ctx[2], found, inv, s0, s1, rects)
    aux = torch.cuda.Stream()

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) / reps

    def serial():
        adam()
        coarse()

    def beside():
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(aux):
            aux.wait_event(ev)
            x = coarse()
            done = torch.cuda.Event()
            done.record()
        adam()
        torch.cuda.current_stream().wait_event(done)
        return x
    print(f"coarse levels alone {timed(coarse):.4f} ms, pass alone {timed(adam):.4f}, one after the other {timed(serial):.4f}, "
          f"beside {timed(beside):.4f}")


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "coarse":
    coarse_beside()
