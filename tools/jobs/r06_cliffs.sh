cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "collisions or per_query or limit" > gpurun_out/r06/gputest_col.log 2>&1; tail -3 gpurun_out/r06/gputest_col.log
timeout 900 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --cliffs > gpurun_out/r06/bench_cliffs_${1:-2}.log 2>&1
python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r06/bench_cliffs_${1:-2}.log') if l.startswith('{\"metric\"')][-1]); print(d['ms_per_step'], d['ms_per_step_p50']); print(json.dumps(d['cliffs'], indent=1))"
