# A/B of the hidden-64 backward variants on one box: usage bash tools/ab_bwd.sh "<flags A>" "<flags B>" ...
cd /root/repo
run() {
  python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s={k[3:]: v for k, v in d['config'].items() if k.startswith('ms_')}; print('$1 | default', round(d['ms_per_step'],3), 'bwd', s['field_bwd'], 'fwd', s['field_fwd'], 'adam', s['adam_coef'])"
  TNL_NO_OVERLAP=1 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s={k[3:]: v for k, v in d['config'].items() if k.startswith('ms_')}; print('$1 | alone  ', round(d['ms_per_step'],3), 'bwd', s['field_bwd'], 'fwd', s['field_fwd'], 'adam', s['adam_coef'])"
}
for flags in "$@"; do
  TNL_HIPCC_FLAGS="$flags" python -m trinerflet_amd.build --force > /dev/null || { echo "build failed: $flags"; continue; }
  run "$flags"; run "$flags"
done
python -m trinerflet_amd.build --force > /dev/null
