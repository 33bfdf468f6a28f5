# Every compile-time knob the kernels keep (an #ifndef default in csrc/): built in its non-default setting(s) and run
# through the parity tests of the kernels it touches.  usage (GPU box): bash tools/knob_ci.sh   (~1 min per line)
# KNOB_CI_DRY=1: print "flags | tests" per line instead of building and testing (tests/test_tools_cpu.py reads that).
cd "$(dirname "$0")/.."
run() {  # flags, test selection
  if [ -n "$KNOB_CI_DRY" ]; then echo "$1 | $2"; return; fi
  TNL_HIPCC_FLAGS="$1" python -m trinerflet_amd.build --force > /dev/null 2>&1 || { echo "BUILD FAILED: $1"; return; }
  echo "[$1] $(python -m pytest $2 -m gpu -x -q 2>&1 | tail -1)"
}
ADAM="tests/test_adam_deferred_gpu.py tests/test_train_gpu.py tests/test_optim_gpu.py"
run "-DTNL_ADAM_PIECE=2048" "$ADAM"
run "-DTNL_ADAM_UNROLL=1" "$ADAM"
run "-DTNL_ADAM_UNROLL=4" "$ADAM"
run "-DTNL_ADAM_ORDER=1" "$ADAM"
run "-DTNL_ADAM_ORDER=2" "$ADAM"
run "-DTNL_ADAM_STORE_ORDER=1" "$ADAM"
run "-DTNL_ADAM_BLOCKS=2048" "$ADAM"
run "-DTNL_ADAM_ZERO_SKIP=0" "$ADAM"
run "-DTNL_RENDER_RT128=128" "tests/test_render_fused_gpu.py"
run "-DTNL_IDWT_BWD_NT=0" "tests/test_idwt_walk_gpu.py tests/test_spans_gpu.py tests/test_roi_gpu.py"
run "-DTNL_FWD_FB=8" "tests/test_idwt_walk_gpu.py tests/test_spans_gpu.py tests/test_roi_gpu.py"
run "-DTNL_ROWS_STAMP=1" "tests/test_field_gpu.py"
# round 5
MARCH="tests/test_raymarching_gpu.py tests/test_full_geometry_gpu.py::test_march_60k_rays_bit_exact"
run "-DTNL_MARCH_WAVE=0" "$MARCH"
run "-DTNL_MARCH_FAST_LANE=0" "$MARCH"
LARGE="tests/test_full_geometry_gpu.py tests/test_field_gpu.py"
run "-DTNL_WG=8 -DTNL_DWG=2" "$LARGE"
run "-DTNL_BWD_STAMP=1" "$LARGE"
run "-DTNL_FWD_PREFETCH=0" "$LARGE"
run "-DTNL_FWD_LDSW=1 -DTNL_FWD_MINWAVES=2" "$LARGE"
# round 5, second half
IDWT="tests/test_idwt_walk_gpu.py tests/test_spans_gpu.py tests/test_roi_gpu.py tests/test_flag_matrix_gpu.py"
run "-DTNL_FWD_PAIR=2" "$IDWT"
run "-DTNL_FWD_PAIR=4" "$IDWT"
run "-DTNL_BWD_WALK_WAVES=3" "$IDWT tests/test_adam_deferred_gpu.py"
run "-DTNL_MARCH_NZ_FILTER=0" "$MARCH tests/test_flag_matrix_gpu.py"
# round 6
run "-DTNL_LAYOUT_ROWS=1" "$IDWT tests/test_triplane_gpu.py"
run "-DTNL_CHAIN_JUMP=0" "$MARCH tests/test_render_fused_gpu.py tests/test_renderer_gpu.py"
run "-DTNL_RENDER_CHAIN_WALK=0" "tests/test_render_fused_gpu.py"
run "-DTNL_SIDE_PRIO=0 -DTNL_MAIN_PRIO=3" "$MARCH $IDWT tests/test_train_gpu.py tests/test_field_gpu.py"
[ -n "$KNOB_CI_DRY" ] || python -m trinerflet_amd.build --force > /dev/null
