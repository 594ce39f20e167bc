#!/bin/bash
o=gpurun_out/r04h; mkdir -p $o
for i in 1 2; do
timeout 200 python bench.py --steps 100 --no-cpu-baseline > $o/bench_default_$i.json 2> $o/bench_default_$i.err < /dev/null
SCCD_EE_SWEEP_EARLY=1 timeout 200 python bench.py --steps 100 --no-cpu-baseline > $o/bench_early_$i.json 2> $o/bench_early_$i.err < /dev/null
done
SCCD_EE_SWEEP_EARLY=1 timeout 600 python -m pytest tests -m gpu -x -q > $o/gputest_early.log 2>&1 < /dev/null; tail -n 2 $o/gputest_early.log
for f in $o/bench_*.json; do echo $f; tail -n 1 $f | cut -c1-140; done
