cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "float" --durations=5 > gpurun_out/r06/gputest_f32cull.log 2>&1; tail -12 gpurun_out/r06/gputest_f32cull.log
timeout 600 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --cliffs 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['cliffs']; print('default', round(d['ms_per_step'],4), 'float', c['float_build_step_ms'], 'collisions', c['collisions_host_path_ms'])"
