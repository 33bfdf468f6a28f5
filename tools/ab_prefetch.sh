# usage (GPU box): bash tools/ab_prefetch.sh [workload] -- the step with the next batch's side work started at three places,
# with the count pass of that march in its two forms (1: one ray per lane, 0: one wavefront per ray)
WL=${1:-base}
for form in 1 0; do
for v in bwd reduce adjoint; do
for rep in 1 2; do
TNL_SIDE_COUNT_FORM=$form TNL_PREFETCH_AT=$v python bench.py --workload $WL --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s={k[3:]: v for k, v in d['config'].items() if k.startswith('ms_')}; print('$WL form $form $v', round(d['ms_per_step'],3), 'periods', s.get('per_step_over_whole_periods'), {k: s[k] for k in ('field_bwd','plane_grad_binned','idwt_adjoint','adam_coef','idwt_fwd','field_fwd','grid_refresh')})"
done
done
done
