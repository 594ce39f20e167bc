# slow steps of a frozen-mesh loop under a few switches: bash tools/jobs/spike_matrix.sh "CFG1 CFG2 ..." [reps] [steps]
for rep in $(seq 1 ${2:-3}); do for cfg in $1; do
  echo -n "$cfg: "; env $cfg python3 tools/jobs/spike_probe.py ${3:-1000} 0 | head -1
done; done
