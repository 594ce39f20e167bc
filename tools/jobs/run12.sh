timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
timeout 300 python bench.py --workload boxes1m --steps 50 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['class_ms_per_step'])"
SCCD_OVERLAP=0 timeout 300 python bench.py --steps 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['class_ms_per_step'])"
timeout 300 python bench.py --steps 50 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['class_ms_per_step'])"
bash tools/timeline.sh cloth1m 2>&1 | tail -40
