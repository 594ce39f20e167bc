# The GPU suite under each non-default laboratory switch that is left (csrc/common.hpp LabEnv), smoke(), and bench.py as 2 / 4 gloo
# ranks sharing the GPU + 1 RCCL rank (functional checks of the multi-process path: the timings of shared-GPU runs mean nothing).
#   bash tools/jobs/env_matrix.sh
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for sw in SCCD_SPECULATE=0 SCCD_SORT_TICKETS=1 SCCD_CULL_SLABS=0 SCCD_NP_WAVES=2 SCCD_SPEC_BREAK=3; do
  echo "== $sw"
  env $sw timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -1
done
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/jobs/multi_rank_functional.sh
