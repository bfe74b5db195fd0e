#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05s11; mkdir -p $O
cd $R
for r in 1 2; do
  for t in base skip1; do
    if [ "$t" = "base" ]; then L=""; else L="--lib $t"; fi
    for G in 3,16,16,2048,19,2 4,10,32,8192,18,4 4,10,16,4096,18,4 3,10,16,8192,20,4; do
      timeout 150 python tools/grid_only.py --iters 8 --grid $G $L 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['lib'], d['grid'], d['bwd_us'], d['grad_abs_sum'])" >> $O/ab.txt
    done
  done
done
cat $O/ab.txt
