#!/bin/bash
o=gpurun_out/r04fix; mkdir -p $o
sha256sum scalable-ccd_amd/sccd/libsccd_hip.so | tee $o/lib.sha256
timeout 100 python3 -m pytest "tests/test_gpu_parity.py::test_memory_limit_halves_the_swept_range" -q -x 2>&1 < /dev/null | tail -n 1
timeout 100 python3 -m pytest tests/test_gpu_parity.py -q -x -k "overflow_behind" 2>&1 < /dev/null | tail -n 1
timeout 200 python3 -m pytest tests -m gpu -q > $o/gputest.log 2>&1 < /dev/null; tail -n 1 $o/gputest.log
SCCD_TEST_ORDER=shuffle:2 timeout 200 python3 -m pytest tests -m gpu -q 2>&1 < /dev/null | tail -n 1
timeout 200 python3 bench.py 2>/dev/null < /dev/null | tail -n 1 > $o/bench_cloth1m.json
python3 -c "
import json; d=json.load(open('$o/bench_cloth1m.json')); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['alone']['frac'], r['traffic'], r['valu_per_check'], r['traffic_note'])"
