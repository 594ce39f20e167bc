# SCCD_EREC_LATE on small meshes: clothball10k and cloths of a few sizes, ms per step with the gate on / off.   bash tools/jobs/erec_small.sh
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for late in 1 0; do
  SCCD_EREC_LATE=$late python3 bench.py --workload clothball10k --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('clothball10k late=$late', round(d['ms_per_step'],4))"
done; done
for n in 100 224 320 500; do
for late in 1 0 1 0; do
  SCCD_EREC_LATE=$late python3 bench.py --cloth-n $n --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('cloth $n late=$late', round(d['ms_per_step'],4))"
done; done
