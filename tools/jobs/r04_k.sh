#!/bin/bash
o=gpurun_out/r04k; mkdir -p $o
timeout 1500 python tools/soak.py 600 300000 > $o/soak600.log 2>&1 < /dev/null; grep -c "FLOAT MISMATCH\|^MISMATCH" $o/soak600.log; grep "SKIP float" $o/soak600.log | wc -l; tail -n 2 $o/soak600.log
grep "MISMATCH" $o/soak600.log | head -5 | cut -c1-300
