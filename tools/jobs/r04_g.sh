#!/bin/bash
o=gpurun_out/r04g; mkdir -p $o
SCCD_SPEC_BREAK=7 timeout 300 python bench.py --jitter 1e-4 --steps 400 > $o/bench_jitter_break7.json 2> $o/bench_jitter_break7.err < /dev/null
SCCD_SPECULATE=0 timeout 300 python bench.py --jitter 1e-4 --steps 200 > $o/bench_jitter_nospec.json 2> $o/bench_jitter_nospec.err < /dev/null
timeout 300 python bench.py --jitter 1e-4 --steps 200 > $o/bench_jitter_1e-4.json 2> $o/bench_jitter_1e-4.err < /dev/null
for f in $o/bench_*.json; do echo $f; tail -n 1 $f | cut -c1-100; done
