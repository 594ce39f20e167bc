# round 6: the final collection (tools/collect_profiles.sh r06) -- everything lands in gpurun_out/prof/
cd "$GRAFT_REPO_ROOT"
bash tools/collect_profiles.sh r06 > gpurun_out/collect_r06.log 2>&1
tail -5 gpurun_out/collect_r06.log
ls gpurun_out/prof | head -80
