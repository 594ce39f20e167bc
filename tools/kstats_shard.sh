# per-kernel device time of the 8 emulated shards of the cloth workload: bash tools/kstats_shard.sh [world]
W=${1:-8}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/kshard && rocprofv3 --kernel-trace --stats -d gpurun_out/kshard --output-format csv -- python3 tools/shard_balance.py --worlds $W --reps 3 --n ${2:-708} > gpurun_out/kshard.log 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/kshard/*/*kernel_stats.csv"))[-1]
rows=list(csv.DictReader(open(f)))
steps=4*$W
tot=0
for r in rows:
    nm=r["Name"]; nm=nm[23:] if nm.startswith("(anon") else nm
    per=float(r["TotalDurationNs"])/1e6/steps
    tot+=per
    print("%-28s calls/step %5.1f  ms/step %.4f  avg_us %.1f" % (nm[:28], int(r["Calls"])/steps, per, float(r["AverageNs"])/1e3))
print("sum ms/rank-step", round(tot,3))
PY
tail -3 gpurun_out/kshard.log
