cd "$GRAFT_REPO_ROOT"; export HSA_ENABLE_IPC_MODE_LEGACY=0
for k in 1 2 3; do
SCCD_FORCE_DIST=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2953$k \
  bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('N=1 rccl', round(d['ms_per_step'],4), d['ms_per_step_p50'], d['device_span_ms']['p50'], d['config']['toi'], d['config']['backend'], d['toi_guess'].get('ms_per_step_global_prior'))"
done
timeout 300 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('no dist', round(d['ms_per_step'],4), d['ms_per_step_p50'], d['device_span_ms']['p50'])"
