# usage (GPU box): bash tools/ab_render_flags.sh "<flags for render.hip>" -- the one-kernel render on a trained field with
# render.hip built with and without extra flags (alternating, 3 rounds); ms per 800 x 800 image at max_steps 4096 per work budget
for rep in 1 2 3; do
  for f in "" "$1"; do
    touch trinerflet_amd/csrc/render.hip; TNL_HIPCC_FILE_FLAGS="render.hip:$f" python -m trinerflet_amd.build > /dev/null 2>&1
    echo "render.hip [$f]: $(PYTHONPATH=. python tools/bench_render_trained.py base 512 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print({k:v["ms_per_image"] for k,v in d["budgets"].items()})')"
  done
done | tee gpurun_out/r06_ab_render_flags.txt
touch trinerflet_amd/csrc/render.hip; python -m trinerflet_amd.build > /dev/null 2>&1
