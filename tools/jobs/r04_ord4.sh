#!/bin/bash
o=gpurun_out/r04ord4; mkdir -p $o
{
echo "# the GPU suite in other test orders (SCCD_TEST_ORDER, tests/conftest.py) on the library WITH the overflow re-sweep fix ($(sha256sum scalable-ccd_amd/sccd/libsccd_hip.so | cut -c1-12)...)"
for ord in shuffle:2 shuffle:3 reverse shuffle:1 shuffle:4; do
  echo "== $ord"
  SCCD_TEST_ORDER=$ord timeout 60 python3 -m pytest tests -m gpu -q 2>&1 < /dev/null | tail -n 1
done
} > $o/gputest_orders.log 2>&1
cat $o/gputest_orders.log
