# usage (GPU box): bash tools/ab_render_walk.sh -- the one-kernel render on a trained field with the empty-cell skip walking
# chains of up to W steps and jumping longer ones (csrc/chain_skip.h), W = 0 (always jump) / 8 / 16 / 1000000 (always walk);
# only render.hip is rebuilt.  Prints ms per 800 x 800 image at max_steps 4096 for the work budgets of bench_render_trained.py.
for W in 0 4 16 1000000 0 4 16; do
  touch trinerflet_amd/csrc/render.hip; TNL_HIPCC_FLAGS="-DTNL_RENDER_CHAIN_WALK=$W" python -m trinerflet_amd.build > /dev/null 2>&1
  echo "walk<=$W: $(PYTHONPATH=. python tools/bench_render_trained.py base 512 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print({k:v["ms_per_image"] for k,v in d["budgets"].items()})')"
done | tee gpurun_out/r06_ab_render_walk.txt
touch trinerflet_amd/csrc/render.hip; python -m trinerflet_amd.build > /dev/null 2>&1
