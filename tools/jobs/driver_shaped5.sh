#!/bin/bash
# the driver's command (fresh process, 20 steps, 5 warm-up) five times, then the 100-step line
out=gpurun_out/${1:-r05a}; mkdir -p $out
for i in 1 2 3 4 5; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench20_$i.json 2> $out/bench20_$i.err
done
python3 bench.py --gpus 1 --steps 100 --warmup 5 > $out/bench100.json 2> $out/bench100.err
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(j["ms_per_step"],4), j.get("ms_per_step_p50"), j.get("ms_per_step_p99"), j.get("ms_first_steps"), j["toi_guess"], j["roofline"]["frac"])
    except Exception as e:
        print(f, "ERR", e)
PY
