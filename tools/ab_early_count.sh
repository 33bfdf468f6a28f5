# usage (GPU box): bash tools/ab_early_count.sh [workloads...] -- TrainStep.early_count (the next batch's count pass beside the field
# forward) against the default, with the count pass in its per-lane / wavefront form and at wave priority 3 / 0; alternating
wl="${@:-base small}"
line() { echo "$1 $2 $(env $3 python bench.py --workload $1 --no-cpu-baseline --no-extras --steps 64 --warmup 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(round(d["ms_per_step"],4), round(c["ms_per_step_over_whole_periods"],4), {k[3:]:round(c[k],3) for k in c if k.startswith("ms_") and not k.startswith("ms_per")})')"; }
for rep in 1 2; do
  for prio in 3 0; do
    touch trinerflet_amd/csrc/raymarch.hip; TNL_HIPCC_FILE_FLAGS="raymarch.hip:-DTNL_COUNT_PRIO=$prio" python -m trinerflet_amd.build > /dev/null 2>&1
    for w in $wl; do
      [ $prio = 3 ] && line $w "default_______________prio3 rep$rep" "TNL_EARLY_COUNT=0"
      line $w "early_lane_form________prio$prio rep$rep" "TNL_EARLY_COUNT=1"
      line $w "early_wave_form________prio$prio rep$rep" "TNL_EARLY_COUNT=1 TNL_SIDE_COUNT_FORM=0"
    done
  done
done | tee gpurun_out/r06_ab_early_count.txt
touch trinerflet_amd/csrc/raymarch.hip; python -m trinerflet_amd.build > /dev/null 2>&1
