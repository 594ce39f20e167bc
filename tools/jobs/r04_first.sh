#!/bin/bash
# round 4, first call: this box's baseline, the fused contract, C3 and the sort, GPU suite
mkdir -p gpurun_out/r04a
python bench.py --steps 100 > gpurun_out/r04a/bench_default.json 2> gpurun_out/r04a/bench_default.err
python bench.py --steps 100 --arith 1 --no-cpu-baseline > gpurun_out/r04a/bench_fma.json 2> gpurun_out/r04a/bench_fma.err
python bench.py --steps 100 --no-cpu-baseline > gpurun_out/r04a/bench_default2.json 2> gpurun_out/r04a/bench_default2.err
python bench.py --workload boxes1m --steps 100 --no-cpu-baseline > gpurun_out/r04a/bench_boxes1m.json 2>&1
python bench.py --workload sort16m --steps 50 --no-cpu-baseline > gpurun_out/r04a/bench_sort16m.json 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04a/gputest.log 2>&1
tail -3 gpurun_out/r04a/gputest.log
cat gpurun_out/r04a/bench_default.json gpurun_out/r04a/bench_fma.json gpurun_out/r04a/bench_default2.json | cut -c1-600
