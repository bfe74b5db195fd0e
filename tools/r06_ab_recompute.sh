#!/bin/bash
# [r6] Same-session A/B of the static step: rgb head backward with recomputed activations (EMER_RGB_RECOMPUTE=1, field_fwd stores geo only)
# against the stored-activation path (=0).  usage (on the GPU box): bash tools/r06_ab_recompute.sh [rounds]
ROUNDS=${1:-3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
for r in $(seq 1 $ROUNDS); do
  for v in 1 2 0; do
    EMER_RGB_RECOMPUTE=$v timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-second-state --no-secondary --no-fp16-state --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']
print('recompute=$v', 'ms/step', round(j['ms_per_step'],4), 'median', round(j['ms_per_step_median'],4), 'p10', round(j['ms_per_step_p10'],4), 'eager', round((j['config']['other_launch_mode'] or {}).get('ms_per_step',0),4), {n.replace('emer_',''): round(v['avg_us'],1) for n,v in k.items() if n in ('emer_field_fwd','emer_rgb_head_bwd_fused','emer_rgb_head_bwd_recompute','emer_neck_bwd_fused','emer_hashgrid_fwd','emer_hashgrid_bwd_params_sliced')})"
  done
done
