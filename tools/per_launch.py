"""Aggregate a rocprofv3 kernel trace per (kernel, grid size): launches, mean / min / total microseconds.

usage: python tools/per_launch.py <dir with *_kernel_trace.csv> [out.csv]
The tables under profiles/*_per_launch.csv are produced by this script.
"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit(f"no *kernel_trace.csv under {root}")
    agg = {}
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
                key = (short(row["Kernel_Name"]), row.get("Grid_Size", row.get("Grid_Size_X", "")))
                a = agg.setdefault(key, [0, 0.0, float("inf")])
                a[0] += 1
                a[1] += dur
                a[2] = min(a[2], dur)
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["kernel", "grid", "launches", "mean_us", "min_us", "total_us"])
    for (k, g), (n, tot, mn) in rows:
        w.writerow([k, g, n, f"{tot / n:.1f}", f"{mn:.1f}", f"{tot:.0f}"])


if __name__ == "__main__":
    main()
