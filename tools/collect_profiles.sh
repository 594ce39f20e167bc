# Refresh profiles/ for the current build (run on the GPU box): bash tools/collect_profiles.sh
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
for W in cloth1m boxes1m sort16m clothball10k; do
  python3 bench.py --workload $W 2>/dev/null | tail -1 > gpurun_out/prof/r01_bench_$W.json.log
done
rm -rf gpurun_out/prof/ks
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/ks --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof/ks.log 2>&1
cp $(ls gpurun_out/prof/ks/*/*kernel_stats.csv | tail -1) gpurun_out/prof/r01_cloth1m_kernel_stats.csv
rm -rf gpurun_out/prof/ks3
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/ks3 --output-format csv -- python3 bench.py --workload boxes1m --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof/ks3.log 2>&1
cp $(ls gpurun_out/prof/ks3/*/*kernel_stats.csv | tail -1) gpurun_out/prof/r01_boxes1m_kernel_stats.csv
rm -rf gpurun_out/prof/ks4
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/ks4 --output-format csv -- python3 bench.py --workload sort16m --steps 5 --warmup 2 > gpurun_out/prof/ks4.log 2>&1
cp $(ls gpurun_out/prof/ks4/*/*kernel_stats.csv | tail -1) gpurun_out/prof/r01_sort16m_kernel_stats.csv
bash tools/pmc_traffic.sh cloth1m > gpurun_out/prof/pmc_traffic_cloth1m.txt 2>&1
cp gpurun_out/pmc_traffic_cloth1m.json gpurun_out/prof/r01_pmc_traffic_cloth1m.json
bash tools/pmc_traffic.sh sort16m > gpurun_out/prof/pmc_traffic_sort16m.txt 2>&1
cp gpurun_out/pmc_traffic_sort16m.json gpurun_out/prof/r01_pmc_traffic_sort16m.json
bash tools/pmc_traffic.sh boxes1m > gpurun_out/prof/pmc_traffic_boxes1m.txt 2>&1
cp gpurun_out/pmc_traffic_boxes1m.json gpurun_out/prof/r01_pmc_traffic_boxes1m.json
bash tools/pmc_sq.sh cloth1m > gpurun_out/prof/pmc_sq_cloth1m.txt 2>&1
cp gpurun_out/pmc_sq_cloth1m.json gpurun_out/prof/r01_pmc_sq_cloth1m.json
python3 tools/shard_balance.py --profile > gpurun_out/prof/r01_shard_balance.log 2>&1
ls -la gpurun_out/prof
