#!/bin/bash
o=gpurun_out/r04w; mkdir -p $o
sha256sum scalable-ccd_amd/sccd/libsccd_hip.so > $o/lib.sha256
timeout 900 python -m pytest tests -m gpu -q > $o/gputest.log 2>&1 < /dev/null; tail -n 3 $o/gputest.log
timeout 600 python tools/soak.py 300 810000 > $o/soak300.log 2>&1 < /dev/null; tail -n 2 $o/soak300.log
timeout 600 python tools/soak_steps.py 300 51000 > $o/soak_steps300.log 2>&1 < /dev/null; tail -n 2 $o/soak_steps300.log
for sw in SCCD_READBACK=copy SCCD_NARROW_ORDER=0 SCCD_SYNC=block; do
  echo "== $sw"
  env $sw timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 < /dev/null | tail -1
done
