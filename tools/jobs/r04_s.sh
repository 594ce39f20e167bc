#!/bin/bash
# the read-back through the gather kernel + mailbox word against copies + event: smoke, quick parity, A/B bench lines
mkdir -p gpurun_out/r04s
O=gpurun_out/r04s
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 < /dev/null; echo "smoke rc=$?"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "speculative or ccd_mesh or cloth or step" > $O/quick.log 2>&1 < /dev/null; tail -n 2 $O/quick.log
for r in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 200 2>/dev/null < /dev/null | tail -n 1 > $O/gather_$r.json
  SCCD_READBACK=copy timeout 300 python bench.py --no-cpu-baseline --steps 200 2>/dev/null < /dev/null | tail -n 1 > $O/copy_$r.json
done
timeout 300 python bench.py --workload boxes1m --no-cpu-baseline 2>/dev/null < /dev/null | tail -n 1 > $O/boxes_gather.json
SCCD_READBACK=copy timeout 300 python bench.py --workload boxes1m --no-cpu-baseline 2>/dev/null < /dev/null | tail -n 1 > $O/boxes_copy.json
timeout 300 python bench.py --workload clothball10k --no-cpu-baseline 2>/dev/null < /dev/null | tail -n 1 > $O/ball_gather.json
SCCD_READBACK=copy timeout 300 python bench.py --workload clothball10k --no-cpu-baseline 2>/dev/null < /dev/null | tail -n 1 > $O/ball_copy.json
for f in $O/*.json; do echo "$f $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d.get('p50_ms'), d.get('p99_ms'))" 2>&1 | tail -n 1)"; done
