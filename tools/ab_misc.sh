cd /root/repo
python -m pytest tests/test_train_gpu.py tests/test_roi_gpu.py tests/test_adam_deferred_gpu.py tests/test_dist_gpu.py -m gpu -q 2>&1 | tail -3
for rep in 1 2; do
python bench.py --workload small --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('small', round(d['ms_per_step'],3))"
TNL_CLIP_FAR=1 python bench.py --workload small --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('small clip_far', round(d['ms_per_step'],3))"
done
python bench.py --no-cpu-baseline > gpurun_out/r03b_bench.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r03b_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac']); print([(e['section'], e['ms_per_step'], round(e['frac'],3)) for e in d['roofline']['top']]); print(d['config']['no_roi_ms_per_step'], d['config']['fp32_planes_ms_per_step'], d['config']['other_workloads']); print(d['config']['trajectory']['wall_ms_per_step'], d['config']['trajectory']['second_half_ms_per_step'], d['config']['inference']['ms_per_image'])"
