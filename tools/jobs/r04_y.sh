#!/bin/bash
mkdir -p gpurun_out/r04y
for sw in SCCD_READBACK=copy SCCD_SYNC=block; do
  echo "== $sw"
  env $sw SCCD_TEST_TIMEOUT=150 timeout 400 python3 -m pytest tests -m gpu -x -q > gpurun_out/r04y/$sw.log 2>&1 < /dev/null
  grep -n "Timeout\|FAILED\|passed\|failed\|test_.*py.*line" gpurun_out/r04y/$sw.log | head -8
done
