#!/bin/bash
# A/B the grid kernels across library variants inside ONE gpurun session (box-to-box variance ~5 %).
# usage: bash tools/ab_grid.sh "<tag> <tag> ..." [rounds] ; every tag except "base" must exist as lib/libemernerf_<tag>.so
TAGS=${1:-base}; ROUNDS=${2:-2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
for r in $(seq 1 $ROUNDS); do
  for t in $TAGS; do
    if [ "$t" = "base" ]; then L=""; else L="--lib $t"; fi
    timeout 150 python $R/tools/grid_only.py --iters 12 $L 2>/dev/null | tail -1
  done
done
