# usage (GPU box): bash tools/ab_small.sh [workload] -- A/B of the side chain's forms on a workload (default small: 16 ch x
# 1024^2, whose dense tail -- 0.8 ms -- is shorter than the prefetched march + tile sort, so the sort's fill pass lands
# beside the field forward: 0.45 ms instead of 0.32).  Alternates: counting sort (default) / one-pass capacity lists
# (TNL_CAPACITY_LISTS=1) / the wavefront-per-ray count pass on the side stream (TNL_SIDE_COUNT_FORM=0).
WL=${1:-small}
run() { echo "$WL $1 rep=$2 $(env $1 python bench.py --workload $WL --no-cpu-baseline --no-extras --steps 64 --warmup 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(round(d["ms_per_step"],4), round(c["ms_per_step_over_whole_periods"],4), {k[3:]:round(c[k],3) for k in c if k.startswith("ms_") and not k.startswith("ms_per")})')"; }
for rep in 1 2 3; do run TNL_CAPACITY_LISTS=0 $rep; run TNL_CAPACITY_LISTS=1 $rep; done | tee gpurun_out/r06_ab_${WL}.txt
