# round 6: the GPU suite (log in gpurun_out/r06/), then a default bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06
tag=${1:-2}
timeout 1800 python3 -m pytest tests -m gpu -q ${SUITE_ARGS:--x} > gpurun_out/r06/gputest_$tag.log 2>&1; echo "gputest rc=$?" >> gpurun_out/r06/gputest_$tag.log
tail -5 gpurun_out/r06/gputest_$tag.log
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06/bench_default_$tag.log 2>&1
tail -c 1500 gpurun_out/r06/bench_default_$tag.log
