#!/bin/bash
o=gpurun_out/r04l; mkdir -p $o
timeout 600 python -m pytest tests -m gpu -x -q > $o/gputest.log 2>&1 < /dev/null; tail -n 2 $o/gputest.log
timeout 900 python tools/soak.py 120 300240 > $o/soak_a.log 2>&1 < /dev/null; grep -c "SKIP float" $o/soak_a.log; tail -n 1 $o/soak_a.log
timeout 900 python tools/soak.py 60 300540 > $o/soak_b.log 2>&1 < /dev/null; grep -c "SKIP float" $o/soak_b.log; tail -n 1 $o/soak_b.log
grep "SKIP float\|MISMATCH" $o/soak_a.log $o/soak_b.log | cut -c1-300 | head
