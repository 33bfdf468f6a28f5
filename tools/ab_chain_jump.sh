# usage (GPU box): bash tools/ab_chain_jump.sh -- the per-lane count pass with the empty-cell skip as chain_skip (default) and
# as the literal add loop (-DTNL_CHAIN_JUMP_COUNT=0, only raymarch.hip is rebuilt), base and small, alternating
line() { echo "$1 $2 rep=$3 $(python bench.py --workload $1 --no-cpu-baseline --no-extras --steps 64 --warmup 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(round(d["ms_per_step"],4), round(c["ms_per_step_over_whole_periods"],4), {k[3:]:round(c[k],3) for k in c if k.startswith("ms_") and not k.startswith("ms_per")})')"; }
for rep in 1 2; do
  touch trinerflet_amd/csrc/raymarch.hip; python -m trinerflet_amd.build > /dev/null 2>&1
  line base jump $rep; line small jump $rep
  touch trinerflet_amd/csrc/raymarch.hip; TNL_HIPCC_FLAGS="-DTNL_CHAIN_JUMP_COUNT=0" python -m trinerflet_amd.build > /dev/null 2>&1
  line base loop $rep; line small loop $rep
done | tee gpurun_out/r06_ab_chain_jump.txt
touch trinerflet_amd/csrc/raymarch.hip; python -m trinerflet_amd.build > /dev/null 2>&1
