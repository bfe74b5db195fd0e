#!/bin/bash
# flow step: pairing in both forwards (base) / in the plain forward only (pj1) / nowhere (nopair16)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for r in 1 2 3; do
  for t in base pj1 nopair16; do
    EMER_LIBSEL_SAME_ABI=1 timeout 400 python tools/ab_bench.py $t --kind flow --rays 2048 --no-extras --no-second-state --no-secondary --no-fp16-state --steps 24 --warmup 6 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']
print('flow@2048 $t', 'ms/step', round(j['ms_per_step'],4), 'median', round(j['ms_per_step_median'],4), 'fwd_jac avg us', round(k.get('emer_hashgrid_fwd_jac',{}).get('avg_us',0),1), 'fwd ms/step', round(k['emer_hashgrid_fwd']['ms_per_step'],4))"
  done
done
