# SQ counters of the sweep kernel in two timing builds (tools/variants.sh, VARIANT_SRC=sweep): two and three waves per SIMD, both with
# the emit's atomic taken away (-DSW_NOATOMIC: private slices, the pair list has holes) on the 1M-box workload:
#   bash tools/jobs/r06_sweep_occ_pmc.sh swN2 swN3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for v in "$@"; do
  export SCCD_LIB=$GRAFT_REPO_ROOT/scalable-ccd_amd/sccd/variants/libsccd_$v.so
  i=0
  for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS" \
           "SQ_INST_LEVEL_VMEM SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_BUSY_CU_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    rm -rf gpurun_out/occ_${v}_$i
    timeout 300 rocprofv3 --pmc $G -d gpurun_out/occ_${v}_$i --output-format csv -- python3 bench.py --workload boxes1m --steps 2 --warmup 1 --clock-warmup 0 --no-cpu-baseline > gpurun_out/occ_${v}_$i.log 2>&1 || tail -3 gpurun_out/occ_${v}_$i.log
  done
  python3 - <<PY
import csv,glob,collections,json
agg=collections.defaultdict(lambda:[0.0,0])
for d in sorted(glob.glob("gpurun_out/occ_${v}_*/")):
    fs=sorted(glob.glob(d+"*/*counter_collection.csv"))
    if not fs: continue
    for r in csv.DictReader(open(fs[-1])):
        if "sweep_band" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]][0]+=float(r["Counter_Value"]); agg[r["Counter_Name"]][1]+=1
out={c:v/n for c,(v,n) in agg.items()}
json.dump({"variant":"$v","workload":"boxes1m","kernel":"sweep_band_k<true, 0>","counters":out}, open("gpurun_out/r06/sweep_occ_pmc_$v.json","w"), indent=1, sort_keys=True)
print("$v"); [print("   %-26s %.5g"%(c,x)) for c,x in sorted(out.items())]
PY
  rm -rf gpurun_out/occ_${v}_*
done
