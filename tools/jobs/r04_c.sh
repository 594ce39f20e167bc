#!/bin/bash
o=gpurun_out/r04c; mkdir -p $o
echo "--- default"; timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "memory_limit" 2>&1 | tail -15 | cut -c1-300
echo "--- radix"; SCCD_SORT=radix timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "memory_limit" 2>&1 | tail -5 | cut -c1-300
echo "--- suite radix"; SCCD_SORT=radix timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$o/prof -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/$o/bench_prof.json 2> $GRAFT_REPO_ROOT/$o/bench_prof.err
cd $GRAFT_REPO_ROOT
f=$(ls $o/prof/*/*kernel_stats.csv $o/prof/*kernel_stats.csv 2>/dev/null | head -1); echo $f; head -30 $f | cut -c1-200
