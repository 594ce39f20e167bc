timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sort_is or overlap_pairs or cell_grid or scan_build or sharded or random_100k or golden or full_size" 2>&1 | tail -3
timeout 300 python bench.py --workload sort16m --steps 20 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-600
timeout 300 python bench.py --workload boxes1m --steps 50 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['class_ms_per_step'])"
SCCD_OVERLAP=0 timeout 300 python bench.py --steps 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['class_ms_per_step'])"
timeout 300 python bench.py --steps 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['class_ms_per_step'])"
bash tools/timeline.sh cloth1m 2>&1 | grep "os_hist\|os_pass\|entry_rec\|cell_fill\|grid_setup\|step"
