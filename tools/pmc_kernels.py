"""Per-kernel means of every counter in a rocprofv3 --pmc output directory (counter_collection.csv files).

usage: python tools/pmc_kernels.py <dir> [substring of the kernel name ...]
Prints one line per (kernel, grid): dispatches and the mean of each counter; with SQ_WAVE_CYCLES present also the
shares WAIT_ANY / WAIT_INST_ANY / ACTIVE_INST_ANY of it (MI355X_MICROARCH.md "rocprofv3 PMC slots")."""
import collections
import csv
import glob
import os
import re
import sys


def main():
    root, keys = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
                if keys and not any(k in name for k in keys):
                    continue
                a = acc[(name[:60], r["Grid_Size"])][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    for (name, grid), cs in sorted(acc.items()):
        means = {c: v[1] / v[0] for c, v in cs.items()}
        n = max(v[0] for v in cs.values())
        line = f"{name:60s} grid {grid:>9s} x{n:<4d} " + "  ".join(f"{c}={m:.4g}" for c, m in sorted(means.items()))
        wc = means.get("SQ_WAVE_CYCLES")
        if wc:
            line += "  |  " + "  ".join(f"{c[3:]}/WAVE_CYCLES={means[c] / wc:.2f}" for c in
                                        ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                                         "SQ_ACTIVE_INST_LDS", "SQ_INST_CYCLES_VMEM") if c in means)
        if "SQ_LDS_BANK_CONFLICT" in means and means.get("SQ_LDS_IDX_ACTIVE"):
            line += f"  LDS conflict share={means['SQ_LDS_BANK_CONFLICT'] / means['SQ_LDS_IDX_ACTIVE']:.2f}"
        print(line)


if __name__ == "__main__":
    main()
