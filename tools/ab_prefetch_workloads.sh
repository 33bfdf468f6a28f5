cd /root/repo
for wl in large small; do
for rep in 1 2; do
for v in start fwd bwd; do
TNL_PREFETCH_AT=$v python bench.py --workload $wl --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl prefetch_at=$v', round(d['ms_per_step'],3))"
done; done; done
