# kernel + memory-copy timeline of one ccd() call from HOST matrices:  bash tools/jobs/host_timeline.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/htl && rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/htl --output-format csv -- python3 tools/host_path.py > gpurun_out/htl.log 2>&1
python3 - <<PY
import csv,glob
kf=sorted(glob.glob("gpurun_out/htl/*/*kernel_trace.csv"))[-1]
mf=sorted(glob.glob("gpurun_out/htl/*/*memory_copy_trace.csv"))
ev=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:30]) for r in csv.DictReader(open(kf))]
if mf:
    for r in csv.DictReader(open(mf[-1])):
        ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")+" "+r.get("Bytes","")))
ev.sort()
starts=[i for i,e in enumerate(ev) if "pack_edges_k" in e[2]]
a=starts[-1]
# the copies of the call precede its first pack kernel: go back to the previous narrow kernel's end
b=a
while b>0 and "np_walk_k" not in ev[b-1][2]: b-=1
t0=ev[b][0]
end=[i for i in range(a,len(ev)) if "np_walk_k<false" in ev[i][2]][0]
for s,e,n in ev[b:end+3]:
    print("%8.1f us  +%7.1f us  %s" % ((s-t0)/1e3,(e-s)/1e3,n))
PY
