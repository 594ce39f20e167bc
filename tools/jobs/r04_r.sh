#!/bin/bash
o=gpurun_out/r04r; mkdir -p $o
for i in 1 2; do
timeout 300 python bench.py --steps 100 > $o/b$i.json 2> $o/b$i.err < /dev/null
tail -n 1 $o/b$i.json | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('run $i', round(d['ms_per_step'],4), 'EE avg launch', r['avg_launch_ms'], 'w/o', d['toi_guess']['ms_per_step_without'], 'clock steps', d['config']['clock_warmup_steps'], d['cpu_baseline']['seconds'])"
done
