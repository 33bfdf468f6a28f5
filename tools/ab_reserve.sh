cd /root/repo
for rep in 1 2 3; do
TNL_PREFETCH_AT=bwd python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['sections_ms']; print('bwd        ', round(d['ms_per_step'],3), {k: s[k] for k in ('field_fwd','field_bwd','plane_grad_binned','idwt_adjoint','adam_coef','idwt_fwd')})"
TNL_ADAM_RESERVE=1 TNL_PREFETCH_AT=bwd python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['sections_ms']; print('bwd+reserve', round(d['ms_per_step'],3), {k: s[k] for k in ('field_fwd','field_bwd','plane_grad_binned','idwt_adjoint','adam_coef','idwt_fwd')})"
done
