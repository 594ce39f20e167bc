cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/prof
timeout 600 python3 bench.py --no-cpu-baseline --max-iter 10000000 2>/dev/null | tail -1 > gpurun_out/prof/r06_bench_cloth1m_max_iter_1e7.json.log
timeout 900 python3 bench.py --no-cpu-baseline --cliffs 2>/dev/null | tail -1 > gpurun_out/prof/r06_bench_cloth1m_cliffs.json.log
timeout 600 python3 bench.py 2>/dev/null | tail -1 > gpurun_out/prof/r06_bench_cloth1m.json.log
python3 -c "
import json
for f in ('max_iter_1e7','cliffs',''):
    n='gpurun_out/prof/r06_bench_cloth1m'+('_'+f if f else '')+'.json.log'
    d=json.loads(open(n).read().strip().splitlines()[-1]); print(f or 'default', round(d['ms_per_step'],4), d['ms_per_step_p50'], d['device_span_ms']['p50'], (d.get('cliffs') or {}).get('max_iter_1e7_step_ms'))
"
