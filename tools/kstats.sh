# per-kernel device time of one bench workload (3 timed + 1 warm-up + 2 profiled steps = 6 steps in the trace): bash tools/kstats.sh <workload> [extra bench args]
W=${1:-cloth1m}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/kstats && timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/kstats --output-format csv -- python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/kstats.log 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/kstats/*/*kernel_stats.csv"))[-1]
rows=list(csv.DictReader(open(f)))
tot=0
for r in rows:
    nm=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
    per=float(r["TotalDurationNs"])/6e6
    tot+=per
    print("%-28s calls/step %5.1f  ms/step %.4f  avg_us %.1f" % (nm[:28], int(r["Calls"])/6, per, float(r["AverageNs"])/1e3))
print("sum ms/step", round(tot,3))
PY
grep '^{"metric"' gpurun_out/kstats.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench ms/step', d['ms_per_step'])"
