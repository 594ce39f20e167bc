# round 6: the driver-shaped line (20 steps) on a box whose host cores are all busy (VERDICT r05 task 1: `stress -c <ncpu>` beside it)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06
N=$(nproc)
echo "nproc=$N" > gpurun_out/r06/stress.log
for k in 1 2 3; do
  timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('quiet  mean %.4f p50 %.4f max %.4f dev_p50 %.4f dev_max %.4f slow %s' % (d['ms_per_step'], d['ms_per_step_p50'], d['ms_per_step_max'], d['device_span_ms']['p50'], d['device_span_ms']['max'], d['slowest_steps'][:2]))" >> gpurun_out/r06/stress.log
done
PIDS=""
for i in $(seq 1 $N); do
  python3 -c "
while True:
    pass" &
  PIDS="$PIDS $!"
done
sleep 2
for k in 1 2 3 4 5; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('loaded mean %.4f p50 %.4f max %.4f dev_p50 %.4f dev_max %.4f slow %s' % (d['ms_per_step'], d['ms_per_step_p50'], d['ms_per_step_max'], d['device_span_ms']['p50'], d['device_span_ms']['max'], d['slowest_steps'][:2]))" >> gpurun_out/r06/stress.log
done
timeout 600 python3 bench.py --steps 2000 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('loaded 2000 steps: mean %.4f p50 %.4f p99 %.4f max %.4f dev_p50 %.4f dev_p99 %.4f dev_max %.4f slow %s' % (d['ms_per_step'], d['ms_per_step_p50'], d['ms_per_step_p99'], d['ms_per_step_max'], d['device_span_ms']['p50'], d['device_span_ms']['p99'], d['device_span_ms']['max'], d['slowest_steps']))" >> gpurun_out/r06/stress.log
kill $PIDS
cat gpurun_out/r06/stress.log
