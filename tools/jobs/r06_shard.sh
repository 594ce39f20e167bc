cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
( echo "== C5 (708): every call from 1"; timeout 600 python3 tools/shard_balance.py --n 708 --reps 5 --worlds 1,2,4,8 --profile 2>&1 | grep -v "^{" 
  echo "== C5 (708): the job's prior (every rank from 0.4592)"; timeout 600 python3 tools/shard_balance.py --n 708 --reps 5 --worlds 1,2,4,8 --global-prior 0.4592 2>&1 | grep "^N="
  echo "== 4M triangles (1416): every call from 1"; timeout 900 python3 tools/shard_balance.py --n 1416 --reps 3 --worlds 1,2,4,8 --profile 2>&1 | grep -v "^{"
  echo "== 4M triangles (1416): the job's prior"; timeout 900 python3 tools/shard_balance.py --n 1416 --reps 3 --worlds 1,8 --global-prior 0.4592 2>&1 | grep "^N="
) > gpurun_out/r06/shard_balance_${1:-1}.log 2>&1
cat gpurun_out/r06/shard_balance_${1:-1}.log
