# quick A/B of one environment switch on the default workload, interleaved: bash tools/jobs/r05_env_ab2.sh VAR "v1 v2" [reps] [steps]
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${3:-3}); do for v in $2; do
  env $1=$v python3 bench.py --steps ${4:-40} --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$1=$v', round(d['ms_per_step'],4), 'p50', round(d['ms_per_step_p50'],4), d['config']['toi'])"
done; done
