#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05s15; mkdir -p $O
cd $R
for r in 1 2 3; do
  for t in base wg128; do
    EMER_LIBSEL_SAME_ABI=1 timeout 300 python tools/ab_bench.py $t --no-extras --no-second-state --no-secondary --no-fp16-state --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']
print('$t', round(j['ms_per_step'],4), 'eager', round((j['config']['other_launch_mode'] or {}).get('ms_per_step',0),4), 'ray_wgrad', round(k['emer_ray_wgrad']['ms_per_step'],4))" >> $O/ab.txt
  done
done
cat $O/ab.txt
