#!/bin/bash
o=gpurun_out/r04f; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/gputest.log 2>&1 < /dev/null; tail -n 3 $o/gputest.log
timeout 200 python bench.py --steps 100 --no-cpu-baseline > $o/bench_default.json 2> $o/bench_default.err < /dev/null
timeout 300 python bench.py --jitter 1e-2 --jitter-fraction 0.1 --jitter-alternate 0.0 --steps 200 > $o/bench_jitter_frac.json 2> $o/bench_jitter_frac.err < /dev/null
timeout 300 python bench.py --jitter 3e-3 --jitter-fraction 0.3 --jitter-alternate 0.0 --steps 200 > $o/bench_jitter_frac2.json 2> $o/bench_jitter_frac2.err < /dev/null
timeout 600 python tools/soak.py 200 92000 > $o/soak200.log 2>&1 < /dev/null; tail -n 2 $o/soak200.log
timeout 600 python tools/soak_steps.py 100 5000 > $o/soak_steps.log 2>&1 < /dev/null; tail -n 2 $o/soak_steps.log
for f in $o/bench_*.json; do echo $f; tail -n 1 $f | cut -c1-200; done
tail -n 3 $o/bench_jitter_frac.err | cut -c1-200
