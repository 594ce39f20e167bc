timeout 1500 python -m pytest tests -x -q -m gpu -k "shard or thousands or full_size" 2>&1 | tail -3
SCCD_LEVEL_BUDGET_MB=1024 timeout 900 python tools/soak.py 100 41000 2>&1 | grep -v "^seed\|amdgpu.ids" | tail -3
python tools/shard_balance.py --profile 2>&1 | grep -v amdgpu | tail -9
bash tools/jobs/tl_shard.sh 2>&1 | tail -42
