# round 6: the 20-step line beside busy host threads -- fewer than the container's CPU quota (competition for cores, no throttling) and more (CFS bandwidth throttling: 100 ms stalls)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06
L=gpurun_out/r06/stress2.log
echo "nproc=$(nproc) cpu.max=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null) loadavg=$(cat /proc/loadavg)" > $L
line() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 mean %.4f p50 %.4f p99 %.4f max %.4f dev_p50 %.4f dev_max %.4f slow %s' % (d['ms_per_step'], d['ms_per_step_p50'], d['ms_per_step_p99'], d['ms_per_step_max'], d['device_span_ms']['p50'], d['device_span_ms']['max'], d['slowest_steps'][:3]))"; }
for H in 0 4 8 12 16 24; do
  PIDS=""
  for i in $(seq 1 $H); do
    python3 -c "
while True:
    pass" &
    PIDS="$PIDS $!"
  done
  sleep 1
  for k in 1 2 3; do timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | line "hogs=$H 20 steps" >> $L; done
  timeout 300 python3 bench.py --steps 1000 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | line "hogs=$H 1000 steps" >> $L
  [ -n "$PIDS" ] && kill $PIDS
  sleep 1
done
cat $L
