# usage (GPU box): bash tools/ab_layout_rows.sh -- the fp16 layout pass (k_to_texel_major_h) with 4 rows per workgroup
# (default) and 1 (-DTNL_LAYOUT_ROWS=1; only wavelet.hip is rebuilt): base and small, alternating; sections in ms
line() { echo "$1 rows=$2 rep=$3 $(python bench.py --workload $1 --no-cpu-baseline --no-extras --steps 64 --warmup 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(round(d["ms_per_step"],4), round(c["ms_per_step_over_whole_periods"],4), {k[3:]:round(c[k],3) for k in c if k.startswith("ms_") and not k.startswith("ms_per")})')"; }
for rep in 1 2; do
  for rows in 4 1 8 2; do
    touch trinerflet_amd/csrc/wavelet.hip; TNL_HIPCC_FLAGS="-DTNL_LAYOUT_ROWS=$rows" python -m trinerflet_amd.build > /dev/null 2>&1
    line base $rows $rep; line small $rows $rep
  done
done | tee gpurun_out/r06_ab_layout_rows.txt
touch trinerflet_amd/csrc/wavelet.hip; python -m trinerflet_amd.build > /dev/null 2>&1
