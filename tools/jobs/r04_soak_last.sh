#!/bin/bash
# more seeds on the final library (pre-scanned on the CPU for oracle run time)
o=gpurun_out/r04soak; mkdir -p $o
sha256sum scalable-ccd_amd/sccd/libsccd_hip.so > $o/lib.sha256; cat $o/lib.sha256
timeout 420 python tools/soak.py 600 910000 > $o/soak_600.log 2>&1 < /dev/null; tail -n 2 $o/soak_600.log
timeout 60 python tools/soak.py 20 900000 > $o/soak_batch_900000.log 2>&1 < /dev/null; tail -n 2 $o/soak_batch_900000.log
timeout 60 python tools/soak.py 20 900520 > $o/soak_batch_900520.log 2>&1 < /dev/null; tail -n 2 $o/soak_batch_900520.log
timeout 300 python tools/soak_steps.py 500 70000 > $o/soak_steps_500.log 2>&1 < /dev/null; tail -n 2 $o/soak_steps_500.log
