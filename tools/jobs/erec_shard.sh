# the records gate on the ranks of a multi-GPU job (emulated in turn on one GPU): the 4M-triangle cloth, gate by the size rule / off / forced
cd $GRAFT_REPO_ROOT
for late in 1 0 2; do
  echo "== SCCD_EREC_LATE=$late"
  SCCD_EREC_LATE=$late python3 tools/shard_balance.py --n ${1:-1416} --reps 3 --worlds ${2:-1,2,4,8} 2>&1 | grep "^N="
done
