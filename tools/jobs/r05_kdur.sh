#!/bin/bash
# average duration of the kernels of the default workload's timed steps, per library variant: bash tools/jobs/r05_kdur.sh variant...
for v in "$@"; do
  L=$GRAFT_REPO_ROOT/scalable-ccd_amd/sccd/variants/libsccd_$v.so
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/kd_$v
  SCCD_LIB=$L rocprofv3 --kernel-trace --stats -d gpurun_out/kd_$v --output-format csv -- python3 bench.py --steps 20 --warmup 3 --clock-warmup 0 --no-cpu-baseline > gpurun_out/kd_$v.log 2>&1
  python3 - <<PY
import csv,glob,json
f=sorted(glob.glob("gpurun_out/kd_$v/*/*kernel_stats.csv"))[-1]
rows={r["Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]:r for r in csv.DictReader(open(f))}
out=["$v"]
for k in ("np_cull_k<true>","np_cull_k<false>","np_walk_k<true, 1, 0>","np_walk_k<false, 1, 0>","sweep_band_k<true, 1>","sweep_band_k<false, 3>"):
    if k in rows: out.append("%s %.1f us x%s"%(k, float(rows[k]["AverageNs"])/1e3, rows[k]["Calls"]))
print(" | ".join(out))
try:
    d=json.loads([l for l in open("gpurun_out/kd_$v.log") if l.startswith('{"metric"')][-1]); print("   bench (traced) ms/step", round(d["ms_per_step"],4), "checks", int(d["config"]["checks_per_step"]))
except Exception as e: print("   no bench line", e)
PY
done
