cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not cull and not slab and not soak_scenes" > gpurun_out/r06/gputest_ab.log 2>&1; tail -3 gpurun_out/r06/gputest_ab.log
for k in 1 2 3; do for f in 1 0; do
  SCCD_EE_FIRST=$f timeout 300 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ee_first=$f mean %.4f p50 %.4f dev_p50 %.4f late %.4f / %.4f guess %.4f' % (d['ms_per_step'], d['ms_per_step_p50'], d['device_span_ms']['p50'], d['late_impact']['ms_per_step_p50'], d['late_impact']['one_launch_ms_per_step_p50'], d['toi_guess']['ms_per_step_with']))"
done; done | tee gpurun_out/r06/ab_first.log
