#!/bin/bash
o=gpurun_out/r04m; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/gputest.log 2>&1 < /dev/null; tail -n 3 $o/gputest.log | cut -c1-300
timeout 300 python bench.py --steps 100 --no-cpu-baseline > $o/bench_default.json 2> $o/bench_default.err < /dev/null
for a in 1e-4 1e-3; do timeout 300 python bench.py --jitter $a --steps 200 > $o/bench_jitter_$a.json 2> $o/bench_jitter_$a.err < /dev/null; done
timeout 300 python bench.py --jitter 1e-3 --jitter-alternate 0.05 --steps 200 > $o/bench_jitter_alt.json 2> $o/bench_jitter_alt.err < /dev/null
timeout 900 python tools/soak_steps.py 300 9000 > $o/soak_steps.log 2>&1 < /dev/null; tail -n 2 $o/soak_steps.log
for f in $o/bench_*.json; do echo $f; tail -n 1 $f | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline())
print(round(d['ms_per_step'],4), d.get('toi_guess'), d.get('toi_guess_hits'), d.get('toi_guess_misses'), d.get('p50_ms'), d.get('p99_ms'), (d.get('roofline') or {}).get('class_ms_per_step'))"; done
