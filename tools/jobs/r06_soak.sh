# round 6 soak: scenes of the old seed range, scenes with a random tolerance (seeds from 2,000,000), step sequences
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
A=${1:-300}; B=${2:-200}; C=${3:-150}
( timeout 2400 python3 tools/soak.py $A 60000 2>/dev/null | grep -v "^$" | tail -n 12
  timeout 2400 python3 tools/soak.py $B 2000000 2>/dev/null | grep -v "^$" | tail -n 12
  timeout 2400 python3 tools/soak_steps.py $C 7000 2>/dev/null | grep -v "^$" | tail -n 8 ) > gpurun_out/r06/soak_${4:-1}.log 2>&1
cat gpurun_out/r06/soak_${4:-1}.log
