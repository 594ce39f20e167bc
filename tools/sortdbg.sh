# timing ablations of os_pass_k (results are invalid for dbg != 0): 1 no global writes, 2 no ranking,
# 4 no look-back, 8 linear writes
for dbg in ${DBGS:-0 1 4 7}; do
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && SCCD_SORT_DBG=$dbg rocprofv3 --kernel-trace --stats -d gpurun_out/prof_dbg$dbg --output-format csv -- python3 bench.py --workload sort16m --steps 3 --warmup 1 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/prof_dbg$dbg/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if "os_" in r["Name"]: print("dbg=$dbg", r["Name"][23:35], "avg_us", round(float(r["AverageNs"])/1e3,1))
PY
done
