# kernel timeline of ONE timed step (start offset, duration, queue): bash tools/timeline.sh [workload]
W=${1:-cloth1m}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/tl && rocprofv3 --kernel-trace -d gpurun_out/tl --output-format csv -- python3 bench.py --workload $W --steps 4 --warmup 2 --clock-warmup 0 --no-cpu-baseline "$@" > gpurun_out/tl.log 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/tl/*/*kernel_trace.csv"))[-1]
rows=list(csv.DictReader(open(f)))
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"],r.get("Queue_Id","?")) for r in rows)
starts=[i for i,e in enumerate(ev) if "vertex_boxes_k" in e[2]]
# (bench.py's steps in order: 2 warm-up, 2 with events on every class, 3 settling, then the 4 timed ones -- from toi = 1 --, then
# steps of other kinds: TL_STEP picks one, default the second timed step)
import os
k=int(os.environ.get("TL_STEP","8"))
a,b=starts[k],starts[k+1]
t0=ev[a][0]
for s,e,n,q in ev[a:b]:
    n=n.replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:28]
    print("%8.1f us  +%7.1f us  q%-3s %s" % ((s-t0)/1e3,(e-s)/1e3,q,n))
print("step %.1f us" % ((ev[b][0]-t0)/1e3))
PY
