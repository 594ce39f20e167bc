#!/bin/bash
# round 4, second call: the new default contract, device-word reduce (1 RCCL rank), moving mesh, GPU suite
o=gpurun_out/r04b; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -x -q > $o/gputest.log 2>&1; tail -3 $o/gputest.log
python bench.py --steps 100 > $o/bench_default.json 2> $o/bench_default.err
python bench.py --steps 100 --arith 0 --no-cpu-baseline > $o/bench_strict.json 2> $o/bench_strict.err
SCCD_FORCE_DIST=1 python bench.py --steps 50 --no-cpu-baseline > $o/bench_rccl1.json 2> $o/bench_rccl1.err
for a in 1e-5 1e-3 3e-2; do python bench.py --jitter $a --steps 200 > $o/bench_jitter_$a.json 2> $o/bench_jitter_$a.err; done
for f in $o/bench_*.json; do echo $f; tail -1 $f | cut -c1-400; done
tail -2 $o/*.err | head -40
