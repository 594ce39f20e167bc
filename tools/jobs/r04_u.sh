#!/bin/bash
mkdir -p gpurun_out/r04u
O=gpurun_out/r04u
for r in 1 2; do
for m in gather copy gatherq gathere; do
  SCCD_READBACK=$m timeout 300 python bench.py --no-cpu-baseline --steps 200 2>/dev/null < /dev/null | tail -n 1 > $O/${m}_$r.json
done
done
SCCD_SYNC=block timeout 300 python bench.py --no-cpu-baseline --steps 200 2>/dev/null < /dev/null | tail -n 1 > $O/block_gather.json
SCCD_SYNC=block SCCD_READBACK=copy timeout 300 python bench.py --no-cpu-baseline --steps 200 2>/dev/null < /dev/null | tail -n 1 > $O/block_copy.json
for f in $O/*.json; do echo "$f $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'])" 2>&1 | tail -n 1)"; done
