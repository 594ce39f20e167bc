# A/B of the helper chain's gates (SCCD_EREC_LATE, SCCD_EFILL_AFTER, SCCD_EBOX_AFTER: kernels of the edge list's build chain ordered
# behind points of the vertex + face chain) with record blocks of 1,024 / 512 / 256 rows (tools/variants.sh: VARIANT_SRC=boxes
# er512="-DER_THREADS_=512" er256="-DER_THREADS_=256" base="").   bash tools/jobs/erec_late_ab.sh "lib:late:fill:box ..." [reps]
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${2:-2}); do
for cfg in $1; do
  IFS=: read v late fill box <<< "$cfg"
  SCCD_EREC_LATE=$late SCCD_EFILL_AFTER=$fill SCCD_EBOX_AFTER=$box SCCD_LIB=$GRAFT_REPO_ROOT/scalable-ccd_amd/sccd/variants/libsccd_$v.so timeout 300 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$cfg', round(d['ms_per_step'],4), 'p50', round(d['ms_per_step_p50'],4), d['config']['toi'])"
done; done
