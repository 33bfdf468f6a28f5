# usage (GPU box): bash tools/ab_bwd_variants.sh <workload> <tag> [tag ...] -- per-kernel time of the field backward for the in-tree
# build ("base") and the experiment builds trinerflet_amd/_variants/lib_<tag>.so (tools/build_variant.py)
WL=$1; shift
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  if [ "$tag" = "base" ]; then unset TNL_LIB_PATH; else export TNL_LIB_PATH=/root/repo/trinerflet_amd/_variants/lib_$tag.so; fi
  rm -rf /tmp/abv; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abv -o v -- python3 /root/repo/tools/bench_field_bwd.py $WL > /tmp/abv.txt 2>&1
  python3 - "$tag" <<PY
import csv, glob, sys
f = glob.glob("/tmp/abv/**/*kernel_stats.csv", recursive=True)[0]
out = []
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "k_field_bwd" in n or "k_slab" in n:
        part = n[n.index("k_field_bwd"):][:40] if "k_field_bwd" in n else "k_slab_reduce"
        out.append(f"{part} {float(r['AverageNs'])/1e3:.1f}us")
print(sys.argv[1], "|", " | ".join(out), "|", open("/tmp/abv.txt").read().strip().splitlines()[-1][:60])
PY
done
