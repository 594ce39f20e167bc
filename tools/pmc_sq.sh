# SQ counters of the narrow-phase kernels (one rocprofv3 --pmc pass per group of counters).  gfx950 exposes no
# SQ_INSTS_CBRANCH(_TAKEN) (profiles/r02_counters_available_gfx950.txt is `rocprofv3 -L` of the box): branch behaviour is
# read from SQ_INSTS_BRANCH, lane utilisation from SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), occupancy from
# 4 x SQ_WAVE_CYCLES / (SIMDs x GRBM_GUI_ACTIVE / 8):
#   bash tools/pmc_sq.sh [workload] [extra bench.py arguments, e.g. --passes-apart]
W=${1:-cloth1m}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH" \
         "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM" \
         "SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS" \
         "SQ_INST_LEVEL_SMEM SQ_WAVES" \
         "SQ_IFETCH SQ_IFETCH_LEVEL" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
         "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM" \
         "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" \
         "SQ_BUSY_CU_CYCLES SQ_WAVES_LT_64 SQ_WAVES_EQ_64" \
         "SQ_INST_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_WAIT_INST_VALU SQ_WAIT_INST_SCA SQ_WAIT_INST_BRANCH SQ_WAIT_INST_MISC" \
         "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcsq_$i
  timeout 600 rocprofv3 --pmc $G -d gpurun_out/pmcsq_$i --output-format csv -- python3 bench.py --workload $W --steps 2 --warmup 1 --clock-warmup 0 --no-cpu-baseline "$@" > gpurun_out/pmcsq_$i.log 2>&1 || tail -3 gpurun_out/pmcsq_$i.log
done
python3 - <<PY
import csv,glob,collections,json
out=collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmcsq_*/")):
    fs=sorted(glob.glob(d+"*/*counter_collection.csv"))
    if not fs: continue
    agg=collections.defaultdict(lambda:[0.0,0])
    for r in csv.DictReader(open(fs[-1])):
        nm=r["Kernel_Name"]
        if not any(k in nm for k in ("np_walk_k", "np_cull_k", "sweep_band", "os_pass_k", "entry_record", "cell_fill_append", "np_level_k")): continue
        nm=nm.replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
        agg[(nm,r["Counter_Name"])][0]+=float(r["Counter_Value"]); agg[(nm,r["Counter_Name"])][1]+=1
    for (k,c),(v,n) in agg.items(): out[k][c]=v/n
import hashlib,os
lib=os.environ.get("SCCD_LIB") or "scalable-ccd_amd/sccd/libsccd_hip.so"
sha=hashlib.sha256(open(lib,"rb").read()).hexdigest()
import sys; sys.path.insert(0, "."); from bench import device_code_sha256; dev=device_code_sha256(lib)  # (the kernels' code objects: bench.py admits a profile by either hash)
json.dump({"workload":"$W","lib_sha256":sha,"device_code_sha256":dev,"note":"rocprofv3 --pmc, one pass per counter group (tools/pmc_sq.sh); per-kernel averages over the launches of a 3-step bench run","kernels":out}, open("gpurun_out/pmc_sq_$W.json","w"), indent=1, sort_keys=True)
for k,v in out.items():
    print(k)
    for c,x in sorted(v.items()): print("   %-26s %.4g"%(c,x))
PY
