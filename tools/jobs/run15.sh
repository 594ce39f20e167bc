SCCD_LEVEL_BUDGET_MB=1024 timeout 900 python tools/soak.py 40 20500 2>&1 | grep -v "^seed\|amdgpu.ids" | tail -8
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
