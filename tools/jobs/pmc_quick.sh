# quick SQ counters of one kernel family: bash tools/jobs/pmc_quick.sh <workload> <kernel substring> [extra bench args]
W=${1:-boxes1m}; K=${2:-sweep_band_k}; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU" \
         "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_SMEM"; do
  i=$((i+1))
  rm -rf gpurun_out/pq_$i
  rocprofv3 --pmc $G -d gpurun_out/pq_$i --output-format csv -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pq_$i.log 2>&1 || tail -3 gpurun_out/pq_$i.log
done
python3 - <<PY
import csv,glob,collections
out=collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pq_*/")):
    fs=sorted(glob.glob(d+"*/*counter_collection.csv"))
    if not fs: continue
    agg=collections.defaultdict(lambda:[0.0,0])
    for r in csv.DictReader(open(fs[-1])):
        nm=r["Kernel_Name"]
        if "$K" not in nm: continue
        nm=nm.replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
        agg[(nm,r["Counter_Name"])][0]+=float(r["Counter_Value"]); agg[(nm,r["Counter_Name"])][1]+=1
    for (k,c),(v,n) in agg.items(): out[k][c]=v/n
for k,v in out.items():
    print(k)
    for c,x in sorted(v.items()): print("   %-26s %.4g"%(c,x))
PY
