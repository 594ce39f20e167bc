# the GPU suite, a 200-scene soak and the profile collection of a build, in one call:  bash tools/jobs/final_check.sh [tag]
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2 > gpurun_out/final_gpu_suite.log
timeout 1200 python3 tools/soak.py 200 70000 2>&1 | grep -E "mismatch|bad|supervisor|FAIL|Error" | tail -15 > gpurun_out/final_soak.log
bash tools/collect_profiles.sh ${1:-r06} > gpurun_out/collect.log 2>&1
