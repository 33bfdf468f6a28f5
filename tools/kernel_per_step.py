"""Per-step total of one kernel from a rocprofv3 kernel trace of bench.py: launches are grouped into steps by the gaps
between them (a new step starts when the previous launch of the kernel ended more than --gap microseconds earlier).

usage: python tools/kernel_per_step.py <dir with *_kernel_trace.csv> <kernel name substring> [out.csv] [gap_us=1500]
Prints, and writes to out.csv, one row per step: index, launches, total microseconds, start (ms since the first launch).
bench.py's phases in order: 4 dry-run steps, 16 set-up steps, W warm-up, K timed, min(K,16) instrumented, and (when the
side work runs under the Adam pass) min(K,16) with it moved away; grid-refresh steps (every 16th) launch other variants.
"""
import csv
import glob
import os
import sys


def main():
    root, key = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else None
    gap = float(sys.argv[4]) if len(sys.argv) > 4 else 1500.0
    f = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if key in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps, last_end = [], None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if last_end is None or (s - last_end) / 1e3 > gap:
            steps.append([0, 0.0, s])
        steps[-1][0] += 1
        steps[-1][1] += (e - s) / 1e3
        last_end = e
    t0 = steps[0][2]
    lines = [("step", "launches", "total_us", "start_ms")]
    for i, (n, tot, s) in enumerate(steps):
        lines.append((i, n, f"{tot:.1f}", f"{(s - t0) / 1e6:.2f}"))
    if out:
        with open(out, "w", newline="") as fh:
            csv.writer(fh).writerows(lines)
    for l in lines:
        print(*l)


if __name__ == "__main__":
    main()
