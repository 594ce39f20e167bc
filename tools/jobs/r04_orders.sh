#!/bin/bash
# the final library once more: the default bench line on another box (twice), the GPU suite in other test orders, more soak
o=gpurun_out/r04orders; mkdir -p $o
sha256sum scalable-ccd_amd/sccd/libsccd_hip.so | tee $o/lib.sha256
for r in 1 2; do timeout 300 python3 bench.py 2>/dev/null < /dev/null | tail -n 1 > $o/bench_cloth1m_$r.json; done
{
echo "# the GPU suite in other test orders (SCCD_TEST_ORDER, tests/conftest.py): the tests share one context; final library of round 4 (3f4730ba...)"
for ord in reverse shuffle:1 shuffle:2 shuffle:3 shuffle:4; do
  echo "== $ord"
  SCCD_TEST_ORDER=$ord timeout 300 python3 -m pytest tests -m gpu -q 2>&1 < /dev/null | tail -n 1
done
echo "== shuffle:5, kernels serialised"
SCCD_TEST_ORDER=shuffle:5 AMD_SERIALIZE_KERNEL=3 timeout 400 python3 -m pytest tests -m gpu -q 2>&1 < /dev/null | tail -n 1
} > $o/gputest_orders.log 2>&1
tail -n 4 $o/gputest_orders.log
timeout 240 python tools/soak.py 300 920000 > $o/soak_300.log 2>&1 < /dev/null; tail -n 2 $o/soak_300.log
for f in $o/bench_*.json; do python3 -c "
import json; d=json.load(open('$f')); r=d['roofline']; print('$f', d['ms_per_step'], d['value'], r['frac'], r['alone']['frac'], r['traffic'], d['cpu_baseline']['value'])"; done
