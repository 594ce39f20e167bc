#!/bin/bash
o=$GRAFT_REPO_ROOT/gpurun_out/r04j; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o col -- python3 $GRAFT_REPO_ROOT/tools/jobs/collisions_probe.py 708 -1 > $o/col.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof2 -o col -- python3 $GRAFT_REPO_ROOT/tools/jobs/collisions_probe.py 708 10000000 > $o/col2.log 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
grep "call" $o/col.log $o/col2.log
for d in prof prof2; do f=$(find $o/$d -name "*kernel_stats.csv" | head -n 1); echo "== $f"; if [ -n "$f" ]; then head -n 16 "$f" | cut -c1-150; fi; done
