timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "overlap_pairs or cell_grid or scan_build or sort_axis or edge_cases or crowded or cursor or memory_limit or sharded or cpu_entry or random_100k or translation or thousands or degenerate or golden or full_size" 2>&1 | tail -3
for ch in 1 2; do
echo chunk $ch
SCCD_SWEEP_CHUNK=$ch timeout 300 python bench.py --workload boxes1m --steps 50 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['class_ms_per_step'])"
SCCD_SWEEP_CHUNK=$ch timeout 300 python bench.py --steps 50 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['broad_phase']['ms_passes_apart'], d['broad_phase']['passes_apart'])"
done
SCCD_SWEEP_DIAG=1 timeout 300 python bench.py --workload boxes1m --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "^\[sweep\]" | tail -1
