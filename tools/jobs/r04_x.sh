#!/bin/bash
for sw in SCCD_NONE=1 SCCD_READBACK=copy SCCD_SYNC=block SCCD_NARROW_ORDER=0; do
  echo "== $sw"
  env $sw timeout 200 python3 -m pytest tests/test_sharding.py -m gpu -x -q 2>&1 < /dev/null | tail -2
done
