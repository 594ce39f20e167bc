#!/bin/bash
mkdir -p gpurun_out/r04v
O=gpurun_out/r04v
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 < /dev/null; echo "smoke rc=$?"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "speculative or ccd_mesh or cloth or step or broad" > $O/quick.log 2>&1 < /dev/null; tail -n 2 $O/quick.log
for r in 1 2; do
for m in gather copy; do for o in 1 0; do
  SCCD_NARROW_ORDER=$o SCCD_READBACK=$m timeout 300 python bench.py --no-cpu-baseline --steps 200 2>/dev/null < /dev/null | tail -n 1 > $O/${m}_order${o}_$r.json
done; done
done
for W in boxes1m clothball10k; do for m in gather copy; do
  SCCD_READBACK=$m timeout 300 python bench.py --workload $W --no-cpu-baseline 2>/dev/null < /dev/null | tail -n 1 > $O/${W}_$m.json
done; done
for f in $O/*.json; do echo "$f $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'])" 2>&1 | tail -n 1)"; done
timeout 600 bash tools/timeline.sh cloth1m > $O/tl_gather.txt 2>&1 < /dev/null
