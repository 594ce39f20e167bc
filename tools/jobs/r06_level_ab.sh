# level-order step (max_iter = 100: below 4,096 -> level order) with variants of narrow.hip:  bash tools/jobs/r06_level_ab.sh name...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
for rep in 1 2; do
for v in "$@"; do
  SCCD_LIB=$GRAFT_REPO_ROOT/scalable-ccd_amd/sccd/variants/libsccd_$v.so timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --max-iter 100 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$v', round(d['ms_per_step'],3), d['config']['toi'], int(d['config']['checks_per_step']))"
done
done | tee gpurun_out/r06/level_ab.log
