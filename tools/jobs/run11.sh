timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "overlap_pairs or cell_grid or scan_build or sharded or random_100k or golden or thousands or translation" 2>&1 | tail -3
SCCD_OVERLAP=0 timeout 300 python bench.py --steps 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['class_ms_per_step'])"
timeout 300 python bench.py --steps 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['class_ms_per_step'])"
bash tools/timeline.sh cloth1m 2>&1 | tail -45
