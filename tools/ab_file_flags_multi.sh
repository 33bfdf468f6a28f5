# usage (GPU box): bash tools/ab_file_flags_multi.sh <tag> <workloads, comma separated> "<spec 1>" "<spec 2>" ... -- the library
# built with each per-object flag spec (build.py TNL_HIPCC_FILE_FLAGS; "" = the default build), two rounds, alternating
tag="$1"; wl=$(echo "$2" | tr ',' ' '); shift; shift
line() { echo "$1 spec=[$2] rep=$3 $(python bench.py --workload $1 --no-cpu-baseline --no-extras --steps 64 --warmup 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(round(d["ms_per_step"],4), round(c["ms_per_step_over_whole_periods"],4), {k[3:]:round(c[k],3) for k in c if k.startswith("ms_") and not k.startswith("ms_per")})')"; }
for rep in $(seq 1 ${REPS:-2}); do
  for f in "" "$@"; do
    TNL_HIPCC_FILE_FLAGS="$f" python -m trinerflet_amd.build --force > /dev/null 2>&1 || echo "BUILD FAILED [$f]"
    for w in $wl; do line $w "$f" $rep; done
  done
done | tee gpurun_out/r06_ab_fileflags_$tag.txt
python -m trinerflet_amd.build --force > /dev/null 2>&1
