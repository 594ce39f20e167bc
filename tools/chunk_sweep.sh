for ch in 24 48 72 96 144 240; do echo "chunk $ch"; SCCD_NP_CHUNK=$ch python tools/shard_balance.py --worlds ${1:-8} --profile --reps 3 2>&1 | grep "rank 0" ; done
