#!/bin/bash
# round-5 session 8: timeline of the default static grid (256 slices per level) and of the flow table after the schedule change; round profile
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05s8; mkdir -p $O
cd $R
timeout 200 python tools/trace_sliced.py --grid 3,10,16,8192,20,4 > $O/trace_static10.txt 2>&1
timeout 200 python tools/trace_sliced.py --grid 4,10,16,4096,18,4 > $O/trace_flowtab.txt 2>&1
tail -22 $O/trace_static10.txt; tail -22 $O/trace_flowtab.txt
bash tools/profile_round.sh r05a > $O/profile_round.log 2>&1
tail -5 $O/profile_round.log
