#!/bin/bash
export SOAK_N=2000 SOAK_SEED=900000 STEPS_N=1000 STEPS_SEED=60000
bash tools/jobs/r04_final.sh
