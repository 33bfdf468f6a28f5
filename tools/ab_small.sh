# usage (GPU box): bash tools/ab_small.sh [workload] -- A/B of the side chain's launch widths on a workload (default small:
# 16 ch x 1024^2, whose dense tail -- 0.8 ms -- is shorter than the prefetched march + tile sort, so the sort's fill pass
# lands beside the field forward: 0.45 ms instead of 0.32).  TNL_SIDE_CAPS="emit,fill" workgroups; 0 = full width.
WL=${1:-small}
run() { echo "$WL caps=$1 rep=$2 $(TNL_SIDE_CAPS=$1 python bench.py --workload $WL --no-cpu-baseline --no-extras --steps 64 --warmup 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(round(d["ms_per_step"],4), round(c["ms_per_step_over_whole_periods"],4), {k[3:]:round(c[k],3) for k in c if k.startswith("ms_") and not k.startswith("ms_per")})')"; }
for rep in 1 2; do for caps in 0,0 0,256 0,512 0,128; do run $caps $rep; done; done | tee gpurun_out/r06_ab_${WL}_caps.txt
