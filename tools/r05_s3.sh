#!/bin/bash
# round-5 session 3: wide pairs / tail split / scheduling block A/B on the xyzt and default static grids; parity of the new default
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05s3; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_a_metric_shape_gpu.py tests/test_kernels_gpu.py -x -q -k "hashgrid or grid" > $O/pytest_grid.log 2>&1; echo "pytest rc $?" >> $O/pytest_grid.log
tail -4 $O/pytest_grid.log
for r in 1 2; do
  for t in r4 wide wt2b32 base wt4b16 wt2b8 wt1b16; do
    if [ "$t" = "base" ]; then L=""; else L="--lib $t"; fi
    timeout 150 python tools/grid_only.py --iters 8 --grid 4,10,32,8192,18,4 $L 2>/dev/null | tail -1 >> $O/ab_xyzt.txt
  done
  for t in r4 base; do
    if [ "$t" = "base" ]; then L=""; else L="--lib $t"; fi
    timeout 150 python tools/grid_only.py --iters 8 --grid 3,10,16,8192,20,4 $L 2>/dev/null | tail -1 >> $O/ab_static10.txt
    timeout 150 python tools/grid_only.py --iters 8 $L 2>/dev/null | tail -1 >> $O/ab_main.txt
  done
done
timeout 200 python tools/trace_sliced.py --grid 4,10,32,8192,18,4 > $O/trace_xyzt_new.txt 2>&1
cut -c1-200 $O/ab_xyzt.txt; cat $O/ab_static10.txt $O/ab_main.txt | cut -c1-200; tail -24 $O/trace_xyzt_new.txt
