#!/bin/bash
# A/B of an environment switch on the default workload: bash tools/jobs/r05_env_ab.sh VAR "v1 v2" [reps]  (ms per step from toi = 1, with the prior, checks)
VAR=$1; VALS=$2; REPS=${3:-2}
for rep in $(seq 1 $REPS); do
for v in $VALS; do
  env $VAR=$v python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); c=d['roofline']['class_ms_per_step']
print('$VAR=$v', round(d['ms_per_step'],4), 'p50', d['ms_per_step_p50'], 'p99', d['ms_per_step_p99'], 'with prior', d['toi_guess']['ms_per_step_with'], 'checks', int(d['config']['checks_per_step']), 'vf', c['narrow_vf'], 'ee', c['narrow_ee'], 'sweep', c['sweep'], d['config']['toi'])"
done
done
