# A/B of the records gate (SCCD_EREC_LATE=0|1|2: the edge list's records kernel ordered behind the END of the vertex + face one) with
# record blocks of 1,024 / 512 / 256 rows (tools/variants.sh: VARIANT_SRC=boxes er1024="-DER_THREADS_=1024" er256="-DER_THREADS_=256" base="").
#   bash tools/jobs/erec_late_ab.sh "base:0 base:2 er1024:0 er1024:2" [reps]      (profiles/r05_ab/build_chain_experiments.txt, item 12)
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${2:-2}); do
for cfg in $1; do
  IFS=: read v late <<< "$cfg"
  SCCD_EREC_LATE=$late SCCD_LIB=$GRAFT_REPO_ROOT/scalable-ccd_amd/sccd/variants/libsccd_$v.so timeout 300 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$cfg', round(d['ms_per_step'],4), 'p50', round(d['ms_per_step_p50'],4), d['config']['toi'])"
done; done
