# interleaved A/B of one environment variable on the default workload: bash tools/jobs/r06_ab_env.sh VAR "v1 v2" [reps] [steps]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
VAR=$1; VALS=$2; REPS=${3:-3}; STEPS=${4:-100}
for k in $(seq 1 $REPS); do for v in $VALS; do
  env $VAR=$v timeout 300 python3 bench.py --steps $STEPS --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v mean %.4f p50 %.4f dev_p50 %.4f late %.4f guess %.4f' % (d['ms_per_step'], d['ms_per_step_p50'], d['device_span_ms']['p50'], d['late_impact']['ms_per_step_p50'], d['toi_guess']['ms_per_step_with']))"
done; done | tee gpurun_out/r06/ab_$VAR.log
