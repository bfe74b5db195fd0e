#!/bin/bash
# round-5 end-of-round profile set (final sources): bench line + kernel stats + stamped HBM traffic (profile_round), grid counters (main,
# dynamic xyzt), head counters, per-config statistics and step sequences, static step sequence
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r05b}
cd $R
bash tools/profile_round.sh $T > gpurun_out/${T}_round.log 2>&1
bash tools/pmc_grid.sh $T > gpurun_out/${T}_grid.log 2>&1
bash tools/pmc_grid.sh ${T}_xyzt --grid 4,10,32,8192,18,4 > gpurun_out/${T}_gridx.log 2>&1
bash tools/pmc_heads.sh $T > gpurun_out/${T}_heads.log 2>&1
bash tools/profile_config.sh $T flow 2048 > gpurun_out/${T}_flow.log 2>&1
bash tools/profile_config.sh $T feature 2048 > gpurun_out/${T}_feat.log 2>&1
bash tools/profile_config.sh $T dynamic 8192 > gpurun_out/${T}_dyn.log 2>&1
bash tools/step_sequence.sh > /dev/null 2>&1; cp gpurun_out/seq/sequence.txt gpurun_out/prof_$T/${T}_step_sequence_static.txt
tail -3 gpurun_out/${T}_round.log; head -c 700 gpurun_out/prof_$T/${T}_bench.json
