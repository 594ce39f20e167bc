timeout 600 python bench.py --steps 100 --warmup 5 2>&1 | tail -1 > gpurun_out/bench_default.json; python -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['ms_per_step'], d['min_toi_latency_ms'], d['broad_phase'], d['roofline']['frac'], d['roofline'].get('executed_frac'), d.get('cpu_baseline',{}).get('value'))"
timeout 600 python bench.py --steps 50 --no-cpu-baseline --max-iter 10000000 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('1e7:', d['ms_per_step'], d['min_toi_latency_ms'])"
timeout 300 python bench.py --workload boxes1m --steps 50 2>&1 | tail -1 | cut -c1-900
timeout 300 python bench.py --workload boxes1m --boxes-variant thin --steps 50 2>&1 | tail -1 | cut -c1-900
timeout 300 python bench.py --workload boxes1m --boxes-n 16000000 --steps 10 2>&1 | tail -1 | cut -c1-900
timeout 300 python bench.py --workload sort16m --steps 20 2>&1 | tail -1 | cut -c1-600
