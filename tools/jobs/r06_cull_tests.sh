cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "cull or slab or condition_4" --durations=15 > gpurun_out/r06/cull_tests.log 2>&1; tail -25 gpurun_out/r06/cull_tests.log
