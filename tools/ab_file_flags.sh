# usage (GPU box): bash tools/ab_file_flags.sh <tag> "<TNL_HIPCC_FILE_FLAGS spec>" [workloads...] -- the library with and without
# per-object extra flags (build.py TNL_HIPCC_FILE_FLAGS), alternating; step and section times in ms.
# e.g. bash tools/ab_file_flags.sh noslp_tile "scatter.hip:-fno-slp-vectorize" base small large
tag="$1"; spec="$2"; shift; shift
wl="${@:-base small}"
line() { echo "$1 spec=[$2] rep=$3 $(python bench.py --workload $1 --no-cpu-baseline --no-extras --steps 64 --warmup 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(round(d["ms_per_step"],4), round(c["ms_per_step_over_whole_periods"],4), {k[3:]:round(c[k],3) for k in c if k.startswith("ms_") and not k.startswith("ms_per")})')"; }
for rep in 1 2; do
  for f in "" "$spec"; do
    TNL_HIPCC_FILE_FLAGS="$f" python -m trinerflet_amd.build --force > /dev/null 2>&1
    for w in $wl; do line $w "$f" $rep; done
  done
done | tee gpurun_out/r06_ab_fileflags_$tag.txt
python -m trinerflet_amd.build --force > /dev/null 2>&1
