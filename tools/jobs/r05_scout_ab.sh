#!/bin/bash
# round 5: the scout (SCCD_OPT_SCOUT = sample queries per lane) on the default workload: ms per step, checks, per-class device time
out=gpurun_out/${1:-r05b}; mkdir -p $out
for rep in 1 2; do
for s in ${SCOUTS:-0 1 2 3}; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --scout $s 2>$out/scout_$s.err | tail -1 > $out/scout_${s}_$rep.json
  python3 - <<PY
import json
d=json.load(open("$out/scout_${s}_$rep.json")); c=d["roofline"]["class_ms_per_step"]
print("scout $s", round(d["ms_per_step"],4), "p50", d["ms_per_step_p50"], "checks", int(d["config"]["checks_per_step"]), "vf", c["narrow_vf"], "ee", c["narrow_ee"], "with prior", d["toi_guess"]["ms_per_step_with"], d["config"]["toi"])
PY
done
done
