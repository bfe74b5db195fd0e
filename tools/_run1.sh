cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r02d > gpurun_out/prof_r02d.log 2>&1
tail -5 gpurun_out/prof_r02d.log | cut -c1-400
