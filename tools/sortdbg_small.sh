# per-kernel time of the sort at a broad-phase size, with the timing ablations of os_pass_k (SCCD_SORT_DBG: results invalid
# unless 0): bash tools/sortdbg_small.sh [n] [bits]      DBGS="0 4" selects the ablations
N=${1:-1700000}; B=${2:-24}
for dbg in ${DBGS:-0 1 2 4 7}; do
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && SCCD_SORT_DBG=$dbg rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sdbg$dbg --output-format csv -- python3 tools/sort_probe.py $N $B 20 > gpurun_out/sdbg$dbg.log 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/prof_sdbg$dbg/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if "os_" in r["Name"]: print("n=$N dbg=$dbg", r["Name"][23:35], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3,1), "min_us", round(float(r["MinNs"])/1e3,1))
PY
done
