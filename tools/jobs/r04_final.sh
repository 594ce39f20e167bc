#!/bin/bash
# the final library of round 4: GPU suite, soaks, env matrix, every profile under profiles/r04_* (collect_profiles.sh)
o=gpurun_out/r04final; mkdir -p $o gpurun_out/prof
sha256sum scalable-ccd_amd/sccd/libsccd_hip.so > $o/lib.sha256
timeout 900 python -m pytest tests -m gpu -q > $o/gputest.log 2>&1 < /dev/null; tail -n 3 $o/gputest.log
timeout 1500 python tools/soak.py ${SOAK_N:-2000} ${SOAK_SEED:-100000} > $o/soak2000.log 2>&1 < /dev/null; tail -n 2 $o/soak2000.log
timeout 900 python tools/soak_steps.py ${STEPS_N:-1000} ${STEPS_SEED:-7000} > $o/soak_steps1000.log 2>&1 < /dev/null; tail -n 2 $o/soak_steps1000.log
timeout 2400 bash tools/collect_profiles.sh r04 > $o/collect.log 2>&1 < /dev/null; tail -n 5 $o/collect.log
timeout 300 python bench.py --steps 100 --no-cpu-baseline --cliffs 2>/dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_cliffs.json.log
timeout 300 python bench.py --steps 100 --no-cpu-baseline --arith 0 2>/dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_strict.json.log
for a in 1e-5 1e-3 3e-3; do timeout 300 python bench.py --jitter $a --steps 200 2>/dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_jitter_$a.json.log; done
timeout 300 python bench.py --jitter 1e-2 --jitter-fraction 0.1 --jitter-alternate 0.0 --steps 200 2>/dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_jitter_mixed.json.log
timeout 300 python bench.py --jitter 1e-3 --jitter-alternate 0.05 --steps 200 2>/dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_jitter_alternating.json.log
SCCD_SPEC_BREAK=7 timeout 300 python bench.py --jitter 1e-4 --steps 400 2>/dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_jitter_1e-4_forced_misses.json.log
SCCD_FORCE_DIST=1 timeout 300 python bench.py --steps 50 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_rccl_1rank.json.log
timeout 1500 bash tools/jobs/env_matrix.sh > gpurun_out/prof/r04_gputest_env_matrix.log 2>&1 < /dev/null; tail -n 12 gpurun_out/prof/r04_gputest_env_matrix.log
ls gpurun_out/prof | head -80
