# A/B of sweep-kernel variants (tools/variants.sh with VARIANT_SRC=sweep) on the box workload and the cloth:  bash tools/jobs/r06_sweep_ab.sh name...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
AB_WORKLOAD=boxes1m AB_REPEAT=2 bash tools/ab.sh "$@"
AB_WORKLOAD=cloth1m AB_REPEAT=2 bash tools/ab.sh "$@"
} 2>&1 | tee gpurun_out/r06/sweep_ab.log
