# The GPU suite under each non-default switch, smoke(), and bench.py as 2 gloo ranks sharing the GPU + 1 RCCL rank
# (functional checks of the multi-process path: the timings of shared-GPU runs mean nothing).   bash tools/jobs/env_matrix.sh
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for sw in SCCD_OVERLAP=0 SCCD_NARROW_BESIDE=0 SCCD_PRESWEEP=0 SCCD_SYNC=block SCCD_SPECULATE=0 SCCD_SORT_TICKETS=1 SCCD_READBACK=copy SCCD_NARROW_ORDER=0 SCCD_EE_EARLY=0 SCCD_SPLIT_BOXES=0 SCCD_CULL_SLABS=0 SCCD_EARLY_VERDICT=0 SCCD_EREC_LATE=0 SCCD_EREC_LATE=2; do
  echo "== $sw"
  env $sw timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -1
done
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for N in 2 4; do
  SCCD_FORCE_DIST=1 SCCD_BENCH_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2950$N \
    bench.py --gpus $N --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('N=$N', d['n_gpus'], round(d['ms_per_step'],3), d['config']['toi'], d['config']['rccl_ranks'], d['config']['backend'], d['scaling'], d.get('rank_max'))"
done
SCCD_FORCE_DIST=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('N=1 rccl', d['n_gpus'], round(d['ms_per_step'],3), d['config']['toi'], d['config']['rccl_ranks'], d['config']['backend'])"
