# Refresh profiles/ for the CURRENT build (run on the GPU box): bash tools/collect_profiles.sh [round tag, default r06]
# Everything lands in gpurun_out/prof/; copy what is to be kept into profiles/ afterwards (the PMC traffic files carry
# the hash of the library they were taken on: bench.py quotes them only for that very build).
R=${1:-r06}
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
for W in cloth1m boxes1m sort16m; do
  rm -rf gpurun_out/prof/ks_$W
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/ks_$W --output-format csv -- python3 bench.py --workload $W --steps 5 --warmup 2 --clock-warmup 0.2 --no-cpu-baseline > gpurun_out/prof/ks_$W.log 2>&1
  cp $(ls gpurun_out/prof/ks_$W/*/*kernel_stats.csv | tail -n 1) gpurun_out/prof/${R}_${W}_kernel_stats.csv
done
# (the default step overlaps two streams: each kernel's own duration, one kernel at a time on the chip, is in this trace)
rm -rf gpurun_out/prof/ks_cloth1m_apart
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/ks_cloth1m_apart --output-format csv -- python3 bench.py --workload cloth1m --steps 5 --warmup 2 --clock-warmup 0.2 --no-cpu-baseline --passes-apart > gpurun_out/prof/ks_cloth1m_apart.log 2>&1
cp $(ls gpurun_out/prof/ks_cloth1m_apart/*/*kernel_stats.csv | tail -1) gpurun_out/prof/${R}_cloth1m_passes_apart_kernel_stats.csv
for W in cloth1m sort16m boxes1m; do
  bash tools/pmc_traffic.sh $W > gpurun_out/prof/pmc_traffic_$W.txt 2>&1
  cp gpurun_out/pmc_traffic_$W.json gpurun_out/prof/${R}_pmc_traffic_$W.json
done
# (the SQ counters belong to one kernel at a time on the chip: the passes apart)
bash tools/pmc_sq.sh cloth1m --passes-apart > gpurun_out/prof/pmc_sq_cloth1m.txt 2>&1
cp gpurun_out/pmc_sq_cloth1m.json gpurun_out/prof/${R}_pmc_sq_cloth1m.json
bash tools/pmc_sq.sh boxes1m > gpurun_out/prof/pmc_sq_boxes1m.txt 2>&1
cp gpurun_out/pmc_sq_boxes1m.json gpurun_out/prof/${R}_pmc_sq_boxes1m.json
timeout 600 python3 tools/shard_balance.py --profile > gpurun_out/prof/${R}_shard_balance.log 2>&1
# the lines bench.py prints once the traffic files are in place (same build: the hashes match)
cp gpurun_out/prof/${R}_pmc_traffic_*.json gpurun_out/prof/${R}_pmc_sq_*.json profiles/ 2>/dev/null
for W in cloth1m boxes1m sort16m clothball10k; do
  timeout 600 python3 bench.py --workload $W 2>/dev/null | tail -1 > gpurun_out/prof/${R}_bench_$W.json.log
done
timeout 600 python3 bench.py --no-cpu-baseline --passes-apart 2>/dev/null | tail -1 > gpurun_out/prof/${R}_bench_cloth1m_passes_apart.json.log
timeout 600 python3 bench.py --no-cpu-baseline --max-iter 10000000 2>/dev/null | tail -1 > gpurun_out/prof/${R}_bench_cloth1m_max_iter_1e7.json.log
# every round: the strict contract, the step without the cull / with one narrow launch per pass, the cliffs, a mesh that moves
timeout 600 python3 bench.py --no-cpu-baseline --arith 0 2>/dev/null | tail -1 > gpurun_out/prof/${R}_bench_cloth1m_strict.json.log
timeout 600 python3 bench.py --no-cpu-baseline --cull 0 2>/dev/null | tail -1 > gpurun_out/prof/${R}_bench_cloth1m_no_cull.json.log
timeout 600 python3 bench.py --no-cpu-baseline --two-halves 0 2>/dev/null | tail -1 > gpurun_out/prof/${R}_bench_cloth1m_one_launch_per_pass.json.log
timeout 900 python3 bench.py --no-cpu-baseline --cliffs 2>/dev/null | tail -1 > gpurun_out/prof/${R}_bench_cloth1m_cliffs.json.log
for J in 1e-4 1e-3 3e-3; do
  timeout 900 python3 bench.py --no-cpu-baseline --jitter $J --steps 1000 2>/dev/null | tail -1 > gpurun_out/prof/${R}_bench_cloth1m_jitter_$J.json.log
done
timeout 900 python3 bench.py --no-cpu-baseline --jitter 1e-3 --jitter-alternate 0.05 --steps 1000 2>/dev/null | tail -1 > gpurun_out/prof/${R}_bench_cloth1m_jitter_alternating.json.log
# the driver's command five times, then its 100-step line
bash tools/jobs/driver_shaped5.sh prof/${R}_driver_shaped > gpurun_out/prof/${R}_driver_shaped_lines.txt 2>&1
timeout 600 python3 bench.py --workload boxes1m --boxes-variant thin 2>/dev/null | tail -1 > gpurun_out/prof/${R}_bench_boxes1m_thin.json.log
timeout 600 python3 bench.py --workload boxes1m --boxes-n 16000000 --steps 10 2>/dev/null | tail -1 > gpurun_out/prof/${R}_bench_boxes16m.json.log
# SURVEY 8d's other C3 shapes: kernel stats and HBM traffic
for V in "thin --boxes-variant thin" "16m --boxes-n 16000000"; do
  set -- $V; T=$1; shift
  rm -rf gpurun_out/prof/ks_boxes_$T
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/ks_boxes_$T --output-format csv -- python3 bench.py --workload boxes1m "$@" --steps 5 --warmup 2 --clock-warmup 0.2 --no-cpu-baseline > gpurun_out/prof/ks_boxes_$T.log 2>&1
  cp $(ls gpurun_out/prof/ks_boxes_$T/*/*kernel_stats.csv | tail -1) gpurun_out/prof/${R}_boxes_${T}_kernel_stats.csv
done
timeout 600 python3 tools/boxes_variants.py > gpurun_out/prof/${R}_boxes_variants.log 2>&1
# (the raw rocprofv3 output is not kept: gpurun merges at most 64 MiB back, and the summaries above are what is judged)
rm -rf gpurun_out/prof/ks_* gpurun_out/pmc_* gpurun_out/pmcsq_* gpurun_out/kstats gpurun_out/kshard
ls -la gpurun_out/prof
