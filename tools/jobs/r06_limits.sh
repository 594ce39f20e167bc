cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "limit or ipc or reads_nothing or golden or randomised or two_halves or speculative" > gpurun_out/r06/gputest_limits.log 2>&1; tail -3 gpurun_out/r06/gputest_limits.log
for k in 1 2; do
timeout 300 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --max-iter 10000000 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('max_iter 1e7: mean %.4f p50 %.4f dev_p50 %.4f waits %.1f toi %r' % (d['ms_per_step'], d['ms_per_step_p50'], d['device_span_ms']['p50'], d['host_waits_per_step'], d['config']['toi']))"
done | tee gpurun_out/r06/limit_bench.log
