"""Timeline of one steady-state training step from a rocprofv3 kernel trace of bench.py.

usage: python tools/step_timeline.py <dir with *_kernel_trace.csv> [anchor kernel substring] [which occurrence from the end | substring of a kernel the step must contain]
Prints every kernel between two consecutive launches of the anchor (default: k_mse_loss), with start offset, duration,
queue (stream) and the idle gap on its queue since the previous kernel ended.
"""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "k_mse_loss"
    sel = sys.argv[3] if len(sys.argv) > 3 else "3"
    f = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
    if sel.isdigit():
        a, b = idx[-int(sel) - 1], idx[-int(sel)]
    else:   # the last step that contains a kernel whose name has `sel` in it (e.g. k_packbits: a grid-refresh step)
        a, b = next((idx[k], idx[k + 1]) for k in range(len(idx) - 2, -1, -1)
                    if any(sel in r["Kernel_Name"] for r in rows[idx[k]:idx[k + 1]]))
    t0 = int(rows[a]["Start_Timestamp"])
    last_end = {}
    busy = 0
    print(f"step wall {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, {b - a} launches")
    for r in rows[a:b]:
        s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{q} gap{gap:7.1f}  {name}")


if __name__ == "__main__":
    main()
