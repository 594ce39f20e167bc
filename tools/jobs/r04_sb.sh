#!/bin/bash
mkdir -p gpurun_out/r04sb
O=gpurun_out/r04sb
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 < /dev/null; echo "smoke rc=$?"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "speculative or ccd_mesh or cloth or step or broad" > $O/quick.log 2>&1 < /dev/null; tail -n 1 $O/quick.log
for r in 1 2 3; do
for m in 1 0; do
  SCCD_SPLIT_BOXES=$m timeout 300 python bench.py --no-cpu-baseline --steps 200 2>/dev/null < /dev/null | tail -n 1 > $O/split${m}_$r.json
done
done
for m in 1 0; do SCCD_SPLIT_BOXES=$m timeout 300 python bench.py --workload clothball10k --no-cpu-baseline 2>/dev/null < /dev/null | tail -n 1 > $O/ball_split${m}.json; done
for f in $O/*.json; do echo "$f $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['roofline'].get('frac'))" 2>&1 | tail -n 1)"; done
timeout 300 bash tools/timeline.sh cloth1m > $O/tl_split.txt 2>&1 < /dev/null
