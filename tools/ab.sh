# A/B of library variants on the GPU box (built by tools/variants.sh): bash tools/ab.sh name1 name2 ...
# prints ms per step and the per-class device times of the 1M-triangle cloth for each variant
# (AB_WORKLOAD=sort16m|boxes1m: that workload's ms per step instead; AB_REPEAT=n: n runs per variant)
W=${AB_WORKLOAD:-cloth1m}
for rep in $(seq 1 ${AB_REPEAT:-1}); do
for v in "$@"; do
  SCCD_LIB=$GRAFT_REPO_ROOT/scalable-ccd_amd/sccd/variants/libsccd_$v.so timeout 300 python3 bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
if '$W'=='cloth1m':
    c=r['class_ms_per_step']; print('$v', round(d['ms_per_step'],4), 'narrow', round(c['narrow_vf']+c['narrow_ee'],4), c, int(d['config']['checks_per_step']), d['config']['toi'])
else:
    print('$v', '$W', round(d['ms_per_step'],4), r.get('class_ms_per_step'), r.get('avg_launch_ms'))"
done
done
