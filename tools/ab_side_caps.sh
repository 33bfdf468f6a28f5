# usage (GPU box): bash tools/ab_side_caps.sh [workloads...] -- workgroup caps of the prefetched march's emit pass and the tile sort's fill
# pass (TrainStep.side_caps, env TNL_SIDE_CAPS="emit,fill"; 0 = uncapped), alternating rounds; step and section times in ms
wl="${@:-base large}"
line() { echo "$1 caps=$2 rep=$3 $(TNL_SIDE_CAPS=$2 python bench.py --workload $1 --no-cpu-baseline --no-extras --steps 64 --warmup 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(round(d["ms_per_step"],4), round(c["ms_per_step_over_whole_periods"],4), {k[3:]:round(c[k],3) for k in c if k.startswith("ms_") and not k.startswith("ms_per")})')"; }
for rep in $(seq 1 ${REPS:-2}); do
  for caps in ${CAPS:-512,256 256,128 1024,512 0,0 512,128 256,256}; do
    for w in $wl; do line $w $caps $rep; done
  done
done | tee gpurun_out/r06_ab_side_caps.txt
