#!/bin/bash
# round-5 session 2: work-item timelines of the owner-computes backward (main grid, xyzt grid), LUT select on the other grids
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05s2; mkdir -p $O
cd $R
timeout 200 python tools/trace_sliced.py > $O/trace_main.txt 2>&1
timeout 200 python tools/trace_sliced.py --grid 4,10,32,8192,18,4 > $O/trace_xyzt.txt 2>&1
for r in 1 2; do
  for t in base lut; do
    if [ "$t" = "base" ]; then L=""; else L="--lib $t"; fi
    for G in 3,10,16,8192,20,4 3,8,16,512,20,1 3,8,16,2048,20,1; do
      timeout 150 python tools/grid_only.py --iters 10 --grid $G $L 2>/dev/null | tail -1 >> $O/ab_lut2.txt
    done
  done
done
tail -30 $O/trace_main.txt; tail -24 $O/trace_xyzt.txt; cat $O/ab_lut2.txt
