# banded exchange (TrainStep.overlap_exchange): its own cost on one GPU, and two gloo ranks on one GPU with / without it
cd /root/repo
for k in 0 2 3 4; do
TNL_EXCHANGE_BANDS=$k python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s={k[3:]: v for k, v in d['config'].items() if k.startswith('ms_')}; print('1 rank, bands $k:', round(d['ms_per_step'],3), 'tile reduction', s['plane_grad_binned'], 'adjoint', s['idwt_adjoint'])"
done
for k in 0 3; do
TNL_EXCHANGE_BANDS=$k python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --same-device --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s={k[3:]: v for k, v in d['config'].items() if k.startswith('ms_')}; print('2 gloo ranks on one GPU, bands $k:', round(d['ms_per_step'],3), {k_: s[k_] for k_ in ('plane_grad_binned','idwt_adjoint','idwt_fwd')})"
done
