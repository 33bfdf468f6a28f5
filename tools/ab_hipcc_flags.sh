# usage (GPU box): bash tools/ab_hipcc_flags.sh "<flags>" [workloads...] -- the whole library built with and without extra
# hipcc flags (TNL_HIPCC_FLAGS, all csrc/*.hip rebuilt), alternating, base / small / large; step and section times in ms.
# e.g. bash tools/ab_hipcc_flags.sh "-fno-slp-vectorize" base small
flags="$1"; shift
wl="${@:-base small}"
tag=$(echo "$flags" | tr -c 'A-Za-z0-9' '_')
line() { echo "$1 flags=[$2] rep=$3 $(python bench.py --workload $1 --no-cpu-baseline --no-extras --steps 64 --warmup 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(round(d["ms_per_step"],4), round(c["ms_per_step_over_whole_periods"],4), {k[3:]:round(c[k],3) for k in c if k.startswith("ms_") and not k.startswith("ms_per")})')"; }
for rep in 1 2; do
  for f in "" "$flags"; do
    TNL_HIPCC_FLAGS="$f" python -m trinerflet_amd.build --force > /dev/null 2>&1
    for w in $wl; do line $w "$f" $rep; done
  done
done | tee gpurun_out/r06_ab_flags_$tag.txt
python -m trinerflet_amd.build --force > /dev/null 2>&1
