"""Instruction mix of one kernel's main loop from `hipcc -S --cuda-device-only` output.

usage: python tools/isa_mix.py file.s <substring of the mangled kernel name>
The main loop is taken as the span of the longest backward branch inside the kernel.
"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.split(":")[0].endswith(key) or
                 (l.startswith("_Z") and key in l.split(":")[0] and ":" in l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start + 1:end]
    labels = {l.split(":")[0].strip(): i for i, l in enumerate(body) if re.match(r"^\.?[A-Za-z_0-9$.]+:", l.strip())}
    best = (0, 0, len(body))
    for i, l in enumerate(body):
        m = re.match(r"\s*s_cbranch_\w+\s+(\S+)", l) or re.match(r"\s*s_branch\s+(\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
            best = (i - labels[m.group(1)], labels[m.group(1)], i)
    loop = body[best[1]:best[2] + 1]
    groups = collections.Counter()
    for l in loop:
        l = l.strip()
        if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
            continue
        op = l.split()[0]
        if op.startswith("v_mfma"): g = "mfma"
        elif op.startswith("v_accvgpr"): g = "v_accvgpr_*"
        elif op.startswith("v_cvt"): g = "v_cvt*"
        elif op.startswith("v_cmp") or op.startswith("v_cndmask"): g = "v_cmp/cndmask"
        elif op.startswith("s_waitcnt"): g = "s_waitcnt"
        elif op.startswith("s_"): g = "salu/branch"
        elif op.startswith("v_mov"): g = "v_mov"
        else: g = op
        groups[g] += 1
    print(f"kernel lines {len(body)}, main loop lines {len(loop)}, instructions {sum(groups.values())}")
    for g, c in groups.most_common(45):
        print(f"{g:30s}{c}")


if __name__ == "__main__":
    main()
