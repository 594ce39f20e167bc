set -x
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "overlap_pairs or cell_grid or scan_build or sort_axis or edge_cases or crowded or cursor or memory_limit or sharded or cpu_entry or random_100k or translation or thousands or degenerate" 2>&1 | tail -15
timeout 300 python bench.py --workload boxes1m --steps 50 --no-cpu-baseline 2>&1 | tail -1
SCCD_OVERLAP=0 timeout 300 python bench.py --steps 30 --no-cpu-baseline 2>&1 | tail -1
timeout 300 python bench.py --steps 30 --no-cpu-baseline 2>&1 | tail -1
