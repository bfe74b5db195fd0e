#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05s13; mkdir -p $O
cd $R
B="--no-cpu-baseline --no-extras --no-second-state --no-secondary --no-fp16-state --steps 24 --warmup 4"
for r in 1 2 3; do
  for v in 1 0; do
    EMER_FUSE_RMLP_WIDE=$v timeout 300 python bench.py --kind feature --rays 2048 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('feature2048 wide=$v', round(d['ms_per_step'],3), round((d['config']['other_launch_mode'] or {}).get('ms_per_step',0),3))" >> $O/ab.txt
  done
done
cat $O/ab.txt
