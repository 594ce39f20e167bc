#!/bin/bash
# the unprofiled bench lines of profiles/r04_* again (a noisy box: p99 spikes of 4 ms in every run of the first collection)
mkdir -p gpurun_out/prof
for W in cloth1m boxes1m sort16m clothball10k; do
  timeout 600 python3 bench.py --workload $W 2>/dev/null | tail -n 1 > gpurun_out/prof/r04_bench_$W.json.log
done
SCCD_OVERLAP=0 timeout 600 python3 bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_passes_apart.json.log
timeout 600 python3 bench.py --no-cpu-baseline --max-iter 10000000 2>/dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_max_iter_1e7.json.log
timeout 600 python3 bench.py --workload boxes1m --boxes-variant thin 2>/dev/null | tail -n 1 > gpurun_out/prof/r04_bench_boxes1m_thin.json.log
timeout 600 python3 bench.py --workload boxes1m --boxes-n 16000000 --steps 10 2>/dev/null | tail -n 1 > gpurun_out/prof/r04_bench_boxes16m.json.log
timeout 300 python bench.py --steps 100 --no-cpu-baseline --cliffs 2>/dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_cliffs.json.log
timeout 300 python bench.py --steps 100 --no-cpu-baseline --arith 0 2>/dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_strict.json.log
for a in 1e-5 1e-4 1e-3 3e-3; do timeout 300 python bench.py --jitter $a --steps 200 2>/dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_jitter_$a.json.log; done
timeout 300 python bench.py --jitter 1e-2 --jitter-fraction 0.1 --jitter-alternate 0.0 --steps 200 2>/dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_jitter_mixed.json.log
timeout 300 python bench.py --jitter 1e-3 --jitter-alternate 0.05 --steps 200 2>/dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_jitter_alternating.json.log
SCCD_SPEC_BREAK=7 timeout 300 python bench.py --jitter 1e-4 --steps 400 2>/dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_jitter_1e-4_forced_misses.json.log
SCCD_FORCE_DIST=1 timeout 300 python bench.py --steps 50 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_rccl_1rank.json.log
timeout 300 python -m pytest tests/test_gpu_parity.py -q -k "explodes or speculative_toi" 2>&1 | tail -n 2
sha256sum scalable-ccd_amd/sccd/libsccd_hip.so
