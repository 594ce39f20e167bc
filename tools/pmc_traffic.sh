# HBM traffic of one bench workload's kernels from the TCC counters, in separate passes as
# MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
W=${1:-cloth1m}; shift   # (further arguments go to bench.py, e.g. --max-iter 100)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$C
  timeout 600 rocprofv3 --pmc $C -d gpurun_out/pmc_$C --output-format csv -- python3 bench.py --workload $W --steps 2 --warmup 1 --clock-warmup 0 --no-cpu-baseline "$@" > gpurun_out/pmc_$C.log 2>&1
done
python3 - <<PY
import csv,glob,collections,json
out=collections.defaultdict(dict)
for C in ("FETCH_SIZE","WRITE_SIZE"):
    f=sorted(glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv"%C))[-1]
    agg=collections.defaultdict(lambda:[0.0,0])
    for r in csv.DictReader(open(f)):
        nm=r["Kernel_Name"]; nm=nm.replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
        agg[nm][0]+=float(r["Counter_Value"]); agg[nm][1]+=1
    for k,(v,n) in agg.items():
        out[k][C+"_KB_per_launch"]=v/n; out[k]["launches"]=n
res={}
for k,v in out.items():
    f=v.get("FETCH_SIZE_KB_per_launch",0.0); w=v.get("WRITE_SIZE_KB_per_launch",0.0)
    # gfx950: FETCH_SIZE reads exactly half the bytes of wide coalesced reads -> x2 (upper bound for narrow accesses)
    res[k]={"fetch_KB_raw":round(f,1),"write_KB":round(w,1),"hbm_bytes_per_launch_corrected":round((2*f+w)*1024),"launches":v["launches"]}
import hashlib,os,sys
lib=os.environ.get("SCCD_LIB") or "scalable-ccd_amd/sccd/libsccd_hip.so"
sha=hashlib.sha256(open(lib,"rb").read()).hexdigest()
import sys; sys.path.insert(0, "."); from bench import device_code_sha256; dev=device_code_sha256(lib)  # (the kernels' code objects: bench.py admits a profile by either hash)
json.dump({"workload":"$W","lib_sha256":sha,"device_code_sha256":dev,"note":"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; read side doubled per MI355X_MICROARCH.md (HBM section)","kernels":res}, open("gpurun_out/pmc_traffic_$W.json","w"), indent=1, sort_keys=True)
for k,v in sorted(res.items(), key=lambda kv:-kv[1]["hbm_bytes_per_launch_corrected"])[:12]: print(k[:40], v)
PY
