#!/bin/bash
# bench lines again with roofline.alone (bench.py only; the library is the one the profiles were taken on)
P=gpurun_out/prof; mkdir -p $P
sha256sum scalable-ccd_amd/sccd/libsccd_hip.so
for W in cloth1m clothball10k; do timeout 300 python3 bench.py --workload $W 2>/dev/null < /dev/null | tail -n 1 > $P/r04_bench_$W.json.log; done
SCCD_OVERLAP=0 timeout 200 python3 bench.py --no-cpu-baseline 2>/dev/null < /dev/null | tail -n 1 > $P/r04_bench_cloth1m_passes_apart.json.log
timeout 200 python3 bench.py --no-cpu-baseline --max-iter 10000000 2>/dev/null < /dev/null | tail -n 1 > $P/r04_bench_cloth1m_max_iter_1e7.json.log
timeout 200 python bench.py --steps 100 --no-cpu-baseline --cliffs 2>/dev/null < /dev/null | tail -n 1 > $P/r04_bench_cloth1m_cliffs.json.log
timeout 200 python bench.py --steps 100 --no-cpu-baseline --arith 0 2>/dev/null < /dev/null | tail -n 1 > $P/r04_bench_cloth1m_strict.json.log
SCCD_FORCE_DIST=1 timeout 200 python bench.py --steps 50 --no-cpu-baseline 2>/dev/null < /dev/null | grep '^{"metric"' | tail -n 1 > $P/r04_bench_cloth1m_rccl_1rank.json.log
timeout 120 python tools/jobs/slow_seed_probe.py 900536 0x1.e461510000000p-3 2>&1 < /dev/null | tail -n 1
for f in $P/r04_bench_cloth1m*.json.log; do python3 -c "
import json,sys; d=json.loads(open('$f').read()); r=d['roofline']; print('$f'.split('bench_')[1], d['ms_per_step'], r.get('frac'), (r.get('alone') or {}).get('frac'), r.get('traffic'))"; done
