#!/bin/bash
# the final library of round 4 (read-back mailbox): GPU suite, every profile under profiles/r04_* (collect_profiles.sh), the extra
# bench lines, the switch matrix, then soaks (seeds pre-scanned on the CPU for oracle run time: tools/jobs/slow_seed_probe.py)
o=gpurun_out/r04final3; mkdir -p $o gpurun_out/prof
sha256sum scalable-ccd_amd/sccd/libsccd_hip.so > $o/lib.sha256
timeout 600 python -m pytest tests -m gpu -q > $o/gputest.log 2>&1 < /dev/null; tail -n 2 $o/gputest.log
timeout 1200 bash tools/collect_profiles.sh r04 > $o/collect.log 2>&1 < /dev/null; tail -n 3 $o/collect.log
timeout 200 python bench.py --steps 100 --no-cpu-baseline --cliffs 2>/dev/null < /dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_cliffs.json.log
timeout 200 python bench.py --steps 100 --no-cpu-baseline --arith 0 2>/dev/null < /dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_strict.json.log
for a in 1e-5 1e-4 1e-3 3e-3; do timeout 200 python bench.py --jitter $a --steps 200 2>/dev/null < /dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_jitter_$a.json.log; done
timeout 200 python bench.py --jitter 1e-2 --jitter-fraction 0.1 --jitter-alternate 0.0 --steps 200 2>/dev/null < /dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_jitter_mixed.json.log
timeout 200 python bench.py --jitter 1e-3 --jitter-alternate 0.05 --steps 200 2>/dev/null < /dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_jitter_alternating.json.log
SCCD_SPEC_BREAK=7 timeout 200 python bench.py --jitter 1e-4 --steps 400 2>/dev/null < /dev/null | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_jitter_1e-4_forced_misses.json.log
SCCD_FORCE_DIST=1 timeout 200 python bench.py --steps 50 --no-cpu-baseline 2>/dev/null < /dev/null | grep '^{"metric"' | tail -n 1 > gpurun_out/prof/r04_bench_cloth1m_rccl_1rank.json.log
timeout 600 bash tools/jobs/env_matrix.sh > gpurun_out/prof/r04_gputest_env_matrix.log 2>&1 < /dev/null; tail -n 14 gpurun_out/prof/r04_gputest_env_matrix.log
timeout 120 python tools/jobs/slow_seed_probe.py 900004 0x1.ecf8ae0000000p-3 > $o/seed_900004.log 2>&1 < /dev/null; tail -n 1 $o/seed_900004.log
timeout 420 python tools/soak_steps.py ${STEPS_N:-500} ${STEPS_SEED:-60000} > $o/soak_steps.log 2>&1 < /dev/null; tail -n 2 $o/soak_steps.log
timeout 420 python tools/soak.py 500 900020 > $o/soak_a.log 2>&1 < /dev/null; tail -n 2 $o/soak_a.log
timeout 420 python tools/soak.py 340 900560 > $o/soak_b.log 2>&1 < /dev/null; tail -n 2 $o/soak_b.log
du -sh gpurun_out
