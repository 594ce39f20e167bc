cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/tls
cat > /tmp/one_rank.py <<'PY'
import os, sys
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "scalable-ccd_amd"))
import sccd
from sccd import scenes
V0, V1, E, F = scenes.folded_cloth(708, seed=7)
ctx = sccd.default_context()
mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
ctx.set_option(sccd.OPT_SHARD_COUNT, 8); ctx.set_option(sccd.OPT_SHARD_RANK, int(os.environ.get("TL_RANK", "3")))
for _ in range(6):
    toi, st = sccd.ccd_mesh(mesh, want_stats=True)
    ctx.synchronize()
print(toi, st)
PY
rocprofv3 --kernel-trace -d gpurun_out/tls --output-format csv -- python3 /tmp/one_rank.py > gpurun_out/tls.log 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/tls/*/*kernel_trace.csv"))[-1]
rows=list(csv.DictReader(open(f)))
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"],r.get("Queue_Id","?")) for r in rows)
starts=[i for i,e in enumerate(ev) if "vertex_boxes_k" in e[2]]
a,b=starts[-2],starts[-1]
t0=ev[a][0]
for s,e,n,q in ev[a:b]:
    n=n.replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:28]
    print("%8.1f us  +%7.1f us  q%-3s %s" % ((s-t0)/1e3,(e-s)/1e3,q,n))
print("step %.1f us" % ((ev[b][0]-t0)/1e3))
PY
