# generic A/B of environment settings on one box: bash tools/ab_env.sh "VAR=a" "VAR=b" ... (alternating, 3 rounds)
cd /root/repo
for rep in 1 2 3; do
for v in "$@"; do
env $v python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s={k[3:]: v for k, v in d['config'].items() if k.startswith('ms_')}; print('$v', round(d['ms_per_step'],3), {k: s[k] for k in ('idwt_fwd','field_fwd','field_bwd','plane_grad_binned','idwt_adjoint','adam_coef','adam_catchup')}, d['roofline']['algorithmic_bytes_per_launch'] if d['roofline']['section']=='adam_coef' else '')"
done; done
