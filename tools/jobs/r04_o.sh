#!/bin/bash
o=gpurun_out/r04o; mkdir -p $o
timeout 200 python tools/jobs/seed_probe_f32.py 500388 2>&1 < /dev/null | tail -4
timeout 900 python -m pytest tests -m gpu -x -q > $o/gputest.log 2>&1 < /dev/null; tail -n 2 $o/gputest.log | cut -c1-200
timeout 1500 python tools/soak.py 300 500300 > $o/soak_a.log 2>&1 < /dev/null; grep -c "SKIP float" $o/soak_a.log; tail -n 1 $o/soak_a.log
timeout 1500 python tools/soak.py 300 300000 > $o/soak_b.log 2>&1 < /dev/null; grep -c "SKIP float" $o/soak_b.log; tail -n 1 $o/soak_b.log
grep "MISMATCH\|TIMEOUT" $o/soak_a.log $o/soak_b.log | head -5 | cut -c1-300
