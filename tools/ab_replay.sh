# usage (GPU box): bash tools/ab_replay.sh -- the replay of deferred optimiser steps (k_adam_l1_catchup) with the no-gradient
# form of the update (default) and with adam1(g_in = 0) (-DTNL_ADAM_REPLAY_FAST=0; only adam.hip is rebuilt): base, large
line() { echo "$1 fast=$2 rep=$3 $(python bench.py --workload $1 --no-cpu-baseline --no-extras --steps 64 --warmup 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(round(d["ms_per_step"],4), round(c["ms_per_step_over_whole_periods"],4), {k[3:]:round(c[k],3) for k in c if k in ("ms_adam_catchup","ms_adam_coef","ms_idwt_fwd","ms_grid_refresh")})')"; }
for rep in 1 2; do for f in 1 0; do
  touch trinerflet_amd/csrc/adam.hip; TNL_HIPCC_FLAGS="-DTNL_ADAM_REPLAY_FAST=$f" python -m trinerflet_amd.build > /dev/null 2>&1
  line base $f $rep; line large $f $rep
done; done | tee gpurun_out/r06_ab_replay.txt
touch trinerflet_amd/csrc/adam.hip; python -m trinerflet_amd.build > /dev/null 2>&1
