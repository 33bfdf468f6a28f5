# usage (GPU box): bash tools/ab_render_flags_bench.sh "<flags for render.hip>" -- bench.py's own trained-field render figure
# (trajectory_figure -> inference_figure: wall clock of model.render per 800 x 800 image) with render.hip built with the
# repository's flags and with extra ones, alternating, 3 rounds
for rep in 1 2 3; do
  for f in "" "$1"; do
    touch trinerflet_amd/csrc/render.hip; TNL_HIPCC_FILE_FLAGS="render.hip:$f" python -m trinerflet_amd.build > /dev/null 2>&1
    echo "render.hip [+$f]: $(python - <<'PY' 2>/dev/null
import torch, bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
rep = bench.trajectory_figure("base", dev)
it = rep["inference_trained"]
print(it["ms_per_image"], "ms per image (one kernel);", it["loop_ms_per_image"], "(loop)")
PY
)"
  done
done | tee gpurun_out/r06_ab_render_flags_bench.txt
touch trinerflet_amd/csrc/render.hip; python -m trinerflet_amd.build > /dev/null 2>&1
