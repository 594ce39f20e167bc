timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_default.json; python -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['ms_per_step'], d['min_toi_latency_ms'], d['broad_phase']['ms_passes_apart'], d['broad_phase']['passes_apart'], d['roofline']['class_ms_per_step'])"
timeout 300 python bench.py --workload boxes1m --steps 50 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['class_ms_per_step'])"
bash tools/timeline.sh cloth1m 2>&1 | tail -38
