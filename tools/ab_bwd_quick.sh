# quick A/B of hipcc flag sets for the field backward on one box: kernel alone (TNL_NO_OVERLAP) and inside the step
# usage: bash tools/ab_bwd_quick.sh "<flags A>" "<flags B>" ...
cd /root/repo
for flags in "$@"; do
  TNL_HIPCC_FLAGS="$flags" python -m trinerflet_amd.build --force > /dev/null || { echo "build failed: $flags"; continue; }
  python -m pytest tests/test_field_gpu.py -m gpu -x -q -k "backward or binned" 2>&1 | tail -1
  TNL_NO_OVERLAP=1 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s={k[3:]: v for k, v in d['config'].items() if k.startswith('ms_')}; print('$flags | alone  ', round(d['ms_per_step'],3), 'bwd', s['field_bwd'], 'fwd', s['field_fwd'], 'adam', s['adam_coef'])"
  python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s={k[3:]: v for k, v in d['config'].items() if k.startswith('ms_')}; print('$flags | default', round(d['ms_per_step'],3), 'bwd', s['field_bwd'], 'fwd', s['field_fwd'], 'adam', s['adam_coef'])"
done
python -m trinerflet_amd.build --force > /dev/null
