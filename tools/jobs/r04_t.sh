#!/bin/bash
# one step's kernel timeline with the gather read-back and with copies
mkdir -p gpurun_out/r04t
timeout 600 bash tools/timeline.sh cloth1m > gpurun_out/r04t/tl_gather.txt 2>&1 < /dev/null
export SCCD_READBACK=copy
timeout 600 bash tools/timeline.sh cloth1m > gpurun_out/r04t/tl_copy.txt 2>&1 < /dev/null
tail -n 3 gpurun_out/r04t/tl_gather.txt gpurun_out/r04t/tl_copy.txt
