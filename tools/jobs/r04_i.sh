#!/bin/bash
o=gpurun_out/r04i; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/gputest.log 2>&1 < /dev/null; tail -n 3 $o/gputest.log
timeout 600 python bench.py --steps 50 --no-cpu-baseline --cliffs > $o/bench_cliffs.json 2> $o/bench_cliffs.err < /dev/null
tail -n 1 $o/bench_cliffs.json | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], json.dumps(d['cliffs'], indent=1))"
tail -n 3 $o/bench_cliffs.err | cut -c1-300
