SCCD_NP_DIAG=2 SCCD_OVERLAP=0 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep "sccd np" | tail -12
