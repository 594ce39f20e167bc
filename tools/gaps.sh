# idle time between consecutive kernels of the timed steps: bash tools/gaps.sh [workload]
# (kernel trace of a short bench run; prints, per pair of neighbouring kernels, the mean gap per step)
W=${1:-cloth1m}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/gaps && rocprofv3 --kernel-trace -d gpurun_out/gaps --output-format csv -- python3 bench.py --workload $W --steps 4 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/gaps.log 2>&1
python3 - <<PY
import csv,glob,collections
f=sorted(glob.glob("gpurun_out/gaps/*/*kernel_trace.csv"))[-1]
rows=list(csv.DictReader(open(f)))
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in rows)
def short(n):
    n=n.replace("(anonymous namespace)::","").replace("void ","")
    return n.split("(")[0][:26]
# steps: from one vertex_boxes_k to the next; keep the last 4 (the timed ones)
starts=[i for i,e in enumerate(ev) if "vertex_boxes_k" in e[2]]
starts=starts[-5:] if len(starts)>=5 else starts
gaps=collections.OrderedDict(); busy=0; total=0; nsteps=0
for a,b in zip(starts[:-1],starts[1:]):
    nsteps+=1
    seg=ev[a:b+1]
    total+=seg[-1][0]-seg[0][0]
    for x,y in zip(seg[:-1],seg[1:]):
        busy+=x[1]-x[0]
        g=y[0]-x[1]
        k=(short(x[2]),short(y[2]))
        gaps.setdefault(k,[0,0]); gaps[k][0]+=g; gaps[k][1]+=1
print("steps",nsteps,"step ms",total/nsteps/1e6,"busy ms",busy/nsteps/1e6)
for k,(g,cnt) in sorted(gaps.items(), key=lambda kv:-kv[1][0])[:25]:
    print("%-26s -> %-26s  %.1f us/step  (%d x %.1f us)" % (k[0],k[1],g/nsteps/1e3,cnt/nsteps,g/cnt/1e3))
PY
grep "^{" gpurun_out/gaps.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench ms/step', d['ms_per_step'])"
