# usage (GPU box): bash tools/ab_side_chain.sh -- round 6's two latency cuts of the prefetched march, each against its
# round-5 form (only raymarch.hip and scatter.hip are rebuilt): the per-lane count pass with the occupancy words in LDS
# (-DTNL_MARCH_LDS_GRID=0: loaded from memory) and the emit pass requesting the next ray's inputs early
# (-DTNL_EMIT_PREFETCH=0), and the tile sort's fill pass likewise (-DTNL_BIN_PREFETCH=0); base, small, large; then the kernel timeline of the default build
python -m pytest tests/test_raymarching_gpu.py "tests/test_full_geometry_gpu.py::test_march_60k_rays_bit_exact" tests/test_edge_cases_gpu.py tests/test_roi_gpu.py -m gpu -q -x 2>&1 | tail -4
line() { echo "$1 [$2] rep=$3 $(python bench.py --workload $1 --no-cpu-baseline --no-extras --steps 64 --warmup 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(round(d["ms_per_step"],4), round(c["ms_per_step_over_whole_periods"],4), {k[3:]:round(c[k],3) for k in c if k.startswith("ms_") and not k.startswith("ms_per")})')"; }
for rep in 1 2; do
  for flags in "" "-DTNL_MARCH_LDS_GRID=0" "-DTNL_EMIT_PREFETCH=0" "-DTNL_BIN_PREFETCH=0" "-DTNL_MARCH_LDS_GRID=0 -DTNL_EMIT_PREFETCH=0 -DTNL_BIN_PREFETCH=0"; do
    touch trinerflet_amd/csrc/raymarch.hip trinerflet_amd/csrc/scatter.hip; TNL_HIPCC_FLAGS="$flags" python -m trinerflet_amd.build > /dev/null 2>&1
    line base "$flags" $rep; line small "$flags" $rep; line large "$flags" $rep
  done
done | tee gpurun_out/r06_ab_side_chain.txt
touch trinerflet_amd/csrc/raymarch.hip trinerflet_amd/csrc/scatter.hip; python -m trinerflet_amd.build > /dev/null 2>&1
bash tools/trace_step.sh base r06_side; bash tools/trace_step.sh small r06_side_small
grep -v "fillBuffer\|elementwise\|k_near_far\|k_step_\|k_scaler\|k_adam_record\|k_mse\|k_slab\|k_field_pack\|copyBuffer" gpurun_out/r06_side_step_timeline.txt | cut -c1-100
