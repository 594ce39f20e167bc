#!/bin/bash
mkdir -p gpurun_out/r04ss
timeout 50 python tools/soak_steps.py 160 80040 > gpurun_out/r04ss/soak_steps_own_ctx_b.log 2>&1 < /dev/null; tail -n 2 gpurun_out/r04ss/soak_steps_own_ctx_b.log
