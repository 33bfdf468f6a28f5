"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately, no trace domains) of bench.py.

usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <out_prefix> <launches_per_step>
writes <out_prefix>_pmc_FETCH_SIZE.csv, <out_prefix>_pmc_WRITE_SIZE.csv (per kernel and grid: dispatches, mean KB per
dispatch, uncorrected) and <out_prefix>_pmc_adam.json: HBM bytes of the step's k_adam_l1 launches over the wavelet
levels + LL, with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts half of a wide streaming read):
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  bench.py reports that figure as roofline.traffic.
"""
import csv
import glob
import json
import os
import re
import sys


def load(root, counter):
    rows = {}
    adam = []
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:70]
                key = (name, r["Grid_Size"])
                a = rows.setdefault(key, [0, 0.0])
                a[0] += 1
                a[1] += float(r["Counter_Value"])
                if "k_adam_l1<true" in r["Kernel_Name"] or "k_adam_l1_live" in r["Kernel_Name"]:   # the step's coefficient launches
                    adam.append(float(r["Counter_Value"]))
    return rows, adam


def main():
    if len(sys.argv) != 5:
        raise SystemExit(__doc__)
    fetch_dir, write_dir, prefix, per_step = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    out = {}
    for counter, root in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        rows, adam = load(root, counter)
        with open(f"{prefix}_pmc_{counter}.csv", "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["kernel", "grid_size", "dispatches", f"avg_{counter}_KB"])
            for (k, g), (n, tot) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
                w.writerow([k, g, n, f"{tot / n:.1f}"])
        steps = len(adam) // per_step
        out[counter] = (sum(adam) / steps, steps)
    f_kb, steps = out["FETCH_SIZE"]
    w_kb, _ = out["WRITE_SIZE"]
    js = {"kernel": f"k_adam_l1_live / k_adam_l1<true>: the {per_step} launches of one step (wavelet levels + LL)",
          "FETCH_SIZE_KB_per_step": f_kb, "WRITE_SIZE_KB_per_step": w_kb, "steps_averaged": steps,
          "hbm_bytes_per_launch": (2 * f_kb + w_kb) * 1024,
          "correction": "gfx950: FETCH_SIZE counts half of a wide coalesced streaming read (MI355X_MICROARCH.md 'HBM'), "
                        "so bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; separate --pmc passes"}
    json.dump(js, open(f"{prefix}_pmc_adam.json", "w"), indent=1)
    print(json.dumps(js))


if __name__ == "__main__":
    main()
