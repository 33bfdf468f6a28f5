# usage (GPU box): bash tools/ab_kernel_times.sh <tag> "<TNL_HIPCC_FILE_FLAGS spec>" <kernel name regex> [workloads...] -- per-kernel
# average durations (rocprofv3 --kernel-trace --stats over bench.py --no-extras --steps 48) with the library built as the
# repository builds it and with the per-object flag spec; prints "workload build kernel calls avg_us" for the matching kernels
tag="$1"; spec="$2"; pat="$3"; shift; shift; shift
wl="${@:-base small}"
R=$(pwd); export TMPDIR=/tmp
for f in "" "$spec"; do
  TNL_HIPCC_FILE_FLAGS="$f" python -m trinerflet_amd.build --force > /dev/null 2>&1
  b=$([ -z "$f" ] && echo default || echo variant)
  for w in $wl; do
    O=/tmp/kt_${tag}_${b}_$w; rm -rf $O
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $R/bench.py --workload $w --steps 48 --warmup 16 --no-cpu-baseline --no-extras > /dev/null 2>&1)
    S=$(find $O -name '*kernel_stats.csv' | head -1)
    python3 - "$S" "$pat" "$w" "$b" <<'PY'
import csv, re, sys
path, pat, w, b = sys.argv[1:5]
for r in csv.DictReader(open(path)):
    if re.search(pat, r["Name"]):
        print(f"{w:6s} {b:8s} {r['Name'][:70]:70s} calls {int(r['Calls']):5d} avg_us {float(r['AverageNs'])/1e3:9.2f}")
PY
    rm -rf $O
  done
done | tee gpurun_out/r06_kt_$tag.txt
python -m trinerflet_amd.build --force > /dev/null 2>&1
