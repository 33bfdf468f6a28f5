"""Paired A/B of one bench.py environment knob: alternating runs, per-pair differences, mean and 95 % CI (Student t).

usage (GPU box): python tools/ab_paired.py NAME A_VALUE B_VALUE [pairs] [workload]     e.g.  TNL_FUSE_LIVE 0 2 6 base
Prints every run's ms_per_step (and over whole periods) and the statistics of B - A."""
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T975 = {1: 12.706, 2: 4.303, 3: 3.182, 4: 2.776, 5: 2.571, 6: 2.447, 7: 2.365, 8: 2.306, 9: 2.262, 10: 2.228, 11: 2.201}


def run(name, value, wl):
    env = dict(os.environ, **{name: str(value)})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--no-extras", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, cwd=ROOT).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    return d["ms_per_step"], d["config"].get("ms_per_step_over_whole_periods")


def main():
    name, a, b = sys.argv[1:4]
    pairs = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    wl = sys.argv[5] if len(sys.argv) > 5 else "base"
    diffs, diffs_p = [], []
    for k in range(pairs):
        order = [(a, 0), (b, 1)] if k % 2 == 0 else [(b, 1), (a, 0)]
        res = {}
        for v, which in order:
            res[which] = run(name, v, wl)
        print(f"pair {k}: {name}={a}: {res[0][0]:.3f} ({res[0][1]})   {name}={b}: {res[1][0]:.3f} ({res[1][1]})", flush=True)
        diffs.append(res[1][0] - res[0][0])
        if res[0][1] is not None and res[1][1] is not None:
            diffs_p.append(res[1][1] - res[0][1])
    for label, ds in (("ms_per_step", diffs), ("over whole periods", diffs_p)):
        if len(ds) < 2:
            continue
        n = len(ds)
        mean = sum(ds) / n
        sd = math.sqrt(sum((x - mean) ** 2 for x in ds) / (n - 1))
        half = T975.get(n - 1, 1.96) * sd / math.sqrt(n)
        print(f"{wl} {label}: {name}={b} minus {name}={a}: mean {mean:+.4f} ms, sd {sd:.4f}, 95% CI [{mean - half:+.4f}, {mean + half:+.4f}] (n={n})")


if __name__ == "__main__":
    main()
