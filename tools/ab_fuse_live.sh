# usage (GPU box): bash tools/ab_fuse_live.sh [workload] -- TrainStep.fuse_live on / off (the optimiser inside the adjoint's
# column-walk levels), alternating repeats
WL=${1:-base}
for rep in 1 2 3; do
for v in 1 0; do
TNL_FUSE_LIVE=$v python bench.py --workload $WL --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s={k[3:]: v for k, v in d['config'].items() if k.startswith('ms_')}; print('$WL fuse_live $v', round(d['ms_per_step'],3), 'periods', s.get('per_step_over_whole_periods'), {k: s[k] for k in ('field_bwd','plane_grad_binned','idwt_adjoint','adam_coef','idwt_fwd','field_fwd','grid_refresh') if k in s})"
done
done
