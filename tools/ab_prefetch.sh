# A/B of where the next batch's march + tile sort start (TrainStep.prefetch_at), alternating runs on one box
cd /root/repo
for rep in 1 2 3 4; do
for v in ${@:-fwd bwd}; do
TNL_PREFETCH_AT=$v python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['sections_ms']; print('prefetch_at=$v', round(d['ms_per_step'],3), {k: s[k] for k in ('idwt_fwd','field_fwd','field_bwd','plane_grad_binned','idwt_adjoint','adam_coef')})"
done; done
