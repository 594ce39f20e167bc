# round 6, first GPU call: the GPU suite, the constructed Condition-4 case on the library WITHOUT the bound, bench default + cliffs
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r06/gputest_1.log 2>&1; echo "gputest rc=$?" >> gpurun_out/r06/gputest_1.log
SCCD_LIB=$GRAFT_REPO_ROOT/scalable-ccd_amd/sccd/variants/libsccd_nocond4.so timeout 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "condition_4" > gpurun_out/r06/cond4_without_bound.log 2>&1
timeout 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_default_1.log 2>&1
timeout 600 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --cliffs > gpurun_out/r06/bench_cliffs_1.log 2>&1
tail -3 gpurun_out/r06/gputest_1.log; tail -5 gpurun_out/r06/cond4_without_bound.log
