#!/bin/bash
o=gpurun_out/r04gdb; mkdir -p $o
T=tests/test_gpu_parity.py::test_memory_limit_halves_the_swept_range
timeout 280 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex run -ex "bt 40" -ex "thread apply all bt 12" --args python3 -m pytest $T -q -x -p no:faulthandler > $o/gdb.log 2>&1 < /dev/null
grep -a -n "SIGABRT\|terminate\|what()\|#[0-9]" $o/gdb.log | head -60
