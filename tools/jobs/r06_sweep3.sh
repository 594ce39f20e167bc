# the sweep at three waves per SIMD: broad-phase tests, then the box and cloth lines:  bash tools/jobs/r06_sweep3.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
timeout 900 python3 -m pytest tests -m gpu -x -q -k "broad or sweep or boxes or pairs" 2>&1 | tail -3
for w in boxes1m boxes1m cloth1m cloth1m; do
timeout 300 python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$w', round(d['ms_per_step'],4), r.get('class_ms_per_step'), r.get('avg_launch_ms'), r.get('frac'))"
done
} 2>&1 | tee gpurun_out/r06/sweep3.log
