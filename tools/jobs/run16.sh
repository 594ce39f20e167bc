timeout 1500 python -m pytest tests -x -q -m gpu -k "shard or thousands or soak" 2>&1 | tail -4
python tools/shard_balance.py --profile 2>&1 | grep -v amdgpu | tail -9
