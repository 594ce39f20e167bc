# kernels of ccd() with the collision list on the 1M-triangle cloth (4 calls):  bash tools/jobs/r06_col_trace.sh [max_iter]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/coltrace
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/coltrace --output-format csv -- python3 tools/jobs/collisions_probe.py 708 ${1:--1} > gpurun_out/coltrace.log 2>&1
tail -4 gpurun_out/coltrace.log
python3 - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/coltrace/*/*kernel_stats.csv"))[-1]
tot=0
for r in list(csv.DictReader(open(f)))[:22]:
    nm=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
    per=float(r["TotalDurationNs"])/4e6; tot+=per
    print("%-34s calls/call %6.1f  ms/call %.4f  avg_us %.1f" % (nm[:34], int(r["Calls"])/4, per, float(r["AverageNs"])/1e3))
print("sum of the listed, ms per call:", round(tot,3))
PY
