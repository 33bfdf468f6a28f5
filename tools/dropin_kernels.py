"""Per-step GPU time of the drop-in loop (tools/bench_dropin.py) by kernel, from a rocprofv3 kernel trace:

    rocprofv3 --kernel-trace -d gpurun_out/dropin -o t --output-format csv -- python3 tools/bench_dropin.py --steps 32
    python tools/dropin_kernels.py gpurun_out/dropin 32

Takes the LAST `steps` steps' worth of the trace (the timed loop: everything after the last launch gap > 20 ms is tear-down
free) and prints, per kernel name, launches and microseconds per step, plus the busy time (union of kernel intervals) and
the span per step: span - busy = the device waiting for the host."""
import csv
import re
import glob
import os
import sys
from collections import defaultdict


def main():
    root, steps = sys.argv[1], int(sys.argv[2])
    f = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    # the optimiser's multi-tensor / fused step is launched once per step: find its launches and cut at the last `steps`
    marks = [s for s, e, n in rows if "k_mse" in n or "mse_loss" in n.lower()]
    if len(marks) < steps + 1:
        marks = [s for s, e, n in rows if "k_composite_train_fwd" in n]
    t_lo, t_hi = marks[-steps - 1], marks[-1]
    sel = [(s, e, n) for s, e, n in rows if t_lo <= s < t_hi]
    per = defaultdict(lambda: [0, 0.0])
    for s, e, n in sel:
        k = re.sub(r"^void ", "", n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", ""))
        k = re.sub(r"at::native::", "", k).split("(")[0][:110]
        per[k][0] += 1
        per[k][1] += (e - s) / 1e3
    busy, cur_s, cur_e = 0.0, None, None
    for s, e, n in sel:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"span {((t_hi - t_lo) / 1e6 / steps):.3f} ms/step, busy {busy / 1e6 / steps:.3f} ms/step, "
          f"{len(sel) / steps:.1f} launches/step")
    for k, (n, us) in sorted(per.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"{us / steps:9.1f} us  {n / steps:6.2f} x  {k}")


if __name__ == "__main__":
    main()
