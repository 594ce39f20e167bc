#!/bin/bash
# ticket-less sort passes: suite + benches, each under a timeout (a dead-locked look-back would hang)
o=gpurun_out/r04e; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/gputest.log 2>&1 < /dev/null; tail -n 3 $o/gputest.log
timeout 200 python bench.py --steps 100 --no-cpu-baseline > $o/bench_default.json 2> $o/bench_default.err < /dev/null
SCCD_SORT_TICKETS=1 timeout 200 python bench.py --steps 100 --no-cpu-baseline > $o/bench_tickets.json 2> $o/bench_tickets.err < /dev/null
timeout 200 python bench.py --steps 100 --no-cpu-baseline > $o/bench_default2.json 2> $o/bench_default2.err < /dev/null
timeout 200 python bench.py --workload boxes1m --steps 100 --no-cpu-baseline > $o/bench_boxes1m.json 2>&1 < /dev/null
SCCD_SORT_TICKETS=1 timeout 200 python bench.py --workload boxes1m --steps 100 --no-cpu-baseline > $o/bench_boxes1m_tickets.json 2>&1 < /dev/null
timeout 200 python bench.py --workload sort16m --steps 50 --no-cpu-baseline > $o/bench_sort16m.json 2>&1 < /dev/null
for a in 1e-3; do timeout 300 python bench.py --jitter $a --jitter-alternate 0.05 --steps 200 > $o/bench_jitter_alt_$a.json 2> $o/bench_jitter_alt_$a.err < /dev/null; done
timeout 300 python bench.py --jitter 1e-4 --jitter-alternate 0.0 --steps 200 > $o/bench_jitter_alt_1e-4.json 2> $o/bench_jitter_alt_1e-4.err < /dev/null
timeout 600 python tools/soak.py 300 91000 > $o/soak300.log 2>&1 < /dev/null; tail -n 3 $o/soak300.log
for f in $o/bench_*.json; do echo $f; tail -n 1 $f | cut -c1-260; done
