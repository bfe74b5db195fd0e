#!/bin/bash
# [r6] Same-session A/B of the xyzt forward with Jacobian store: encoding stored before the Jacobian (base) vs the round-4 instruction order
# (libemernerf_jacold.so = -DEMER_JAC_ENC_FIRST=0).  Flow step at the 2048-ray shard and at 8192 rays.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for r in 1 2 3; do
  for t in base jacold; do
    for rays in 2048 8192; do
      EMER_LIBSEL_SAME_ABI=1 timeout 400 python tools/ab_bench.py $t --kind flow --rays $rays --no-extras --no-second-state --no-secondary --no-fp16-state --steps 24 --warmup 6 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']; rx=j.get('roofline_xyzt') or {}
print('flow@$rays $t', 'ms/step', round(j['ms_per_step'],4), 'median', round(j['ms_per_step_median'],4), 'fwd_jac avg us', round(k.get('emer_hashgrid_fwd_jac',{}).get('avg_us',0),1), 'x', k.get('emer_hashgrid_fwd_jac',{}).get('launches_per_step'), 'fwd avg us', round(k.get('emer_hashgrid_fwd',{}).get('avg_us',0),1))"
    done
  done
done
