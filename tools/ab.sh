# A/B of library variants on the GPU box (built by tools/variants.sh): bash tools/ab.sh name1 name2 ...
# prints ms per step and the per-class device times of the 1M-triangle cloth for each variant
for v in "$@"; do
  SCCD_LIB=$GRAFT_REPO_ROOT/scalable-ccd_amd/sccd/variants/libsccd_$v.so timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); c=d['roofline']['class_ms_per_step']; print('$v', round(d['ms_per_step'],4), 'narrow', round(c['narrow_vf']+c['narrow_ee'],4), c, int(d['config']['checks_per_step']), d['config']['toi'])"
done
