#!/bin/bash
# every command under its own timeout; nothing reads stdin
o=gpurun_out/r04d; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/gputest.log 2>&1 < /dev/null; tail -n 3 $o/gputest.log
timeout 300 python bench.py --steps 100 > $o/bench_default.json 2> $o/bench_default.err < /dev/null
SCCD_LAZY_ONE=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline > $o/bench_lazy.json 2> $o/bench_lazy.err < /dev/null
timeout 300 python bench.py --steps 100 --arith 0 --no-cpu-baseline > $o/bench_strict.json 2> $o/bench_strict.err < /dev/null
SCCD_FORCE_DIST=1 timeout 300 python bench.py --steps 50 --no-cpu-baseline > $o/bench_rccl1.json 2> $o/bench_rccl1.err < /dev/null
for a in 1e-5 1e-4 1e-3 3e-3; do timeout 300 python bench.py --jitter $a --steps 200 > $o/bench_jitter_$a.json 2> $o/bench_jitter_$a.err < /dev/null; done
for f in $o/bench_*.json; do echo $f; tail -n 1 $f | cut -c1-300; done
tail -n 3 $o/bench_rccl1.err | cut -c1-300
