# bench.py as 2 / 4 gloo ranks sharing the GPU and as 1 RCCL rank (functional: the timings of shared-GPU runs mean nothing).   bash tools/jobs/multi_rank_functional.sh
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in 2 4; do
  SCCD_FORCE_DIST=1 SCCD_BENCH_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N \
    bench.py --gpus $N --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('N=$N', d['n_gpus'], round(d['ms_per_step'],3), d['config']['toi'], d['config']['rccl_ranks'], d['config']['backend'], d['scaling'], d.get('rank_max'))"
done
SCCD_FORCE_DIST=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 \
  bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('N=1 rccl', d['n_gpus'], round(d['ms_per_step'],3), d['config']['toi'], d['config']['rccl_ranks'], d['config']['backend'], d['toi_guess'].get('ms_per_step_global_prior'))"
