# the records gate (SCCD_EREC_LATE) over mesh sizes: ms per step with the gate forced (2) / off (0), interleaved.
#   bash tools/jobs/erec_small.sh "100 224 320 400 500" [reps]       (cloth sides; the size rule of drivers.hip came from this)
cd $GRAFT_REPO_ROOT
for n in ${1:-100 224 320 400 500}; do
for rep in $(seq 1 ${2:-3}); do
for late in 2 0; do
  SCCD_EREC_LATE=$late python3 bench.py --cloth-n $n --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('cloth $n late=$late', round(d['ms_per_step'],4), 'p50', d['ms_per_step_p50'])"
done; done; done
