SCCD_SWEEP_DIAG=1 SCCD_OVERLAP=0 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "^\[sweep\]" | tail -4
SCCD_SWEEP_DIAG=1 timeout 300 python bench.py --workload boxes1m --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "^\[sweep\]" | tail -1
