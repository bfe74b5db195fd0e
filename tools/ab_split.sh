#!/bin/bash
# Same-session A/B of the level-split table backward under the data-parallel exchange with ONE rank (EMER_DP_FORCE=1: real RCCL
# collectives, each a copy onto itself): what the split + the extra collective cost in kernel time and launches.
# usage: bash tools/ab_split.sh [rounds]
R=${GRAFT_REPO_ROOT:-$(pwd)}; ROUNDS=${1:-2}
for r in $(seq 1 $ROUNDS); do
  for sp in 1 0; do
    EMER_DP_FORCE=1 EMER_DP_SPLIT_TABLE=$sp timeout 200 python $R/bench.py --gpus 1 --steps 60 --warmup 10 --no-extras --no-secondary --no-cpu-baseline --no-second-state --no-fp16-state 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=j['gradient_exchange']
print('split=$sp eager', round(j['ms_per_step'],4), 'graph', round((j['config']['other_launch_mode'] or {}).get('ms_per_step',0),4), 'exposed_comm_ms', round(g['exposed_comm_ms'],4), 'table bwd us', round(j['roofline']['grid_encode_plus_bwd']['bwd_avg_us'],1), {n: (round(v['launches_per_step'],2), round(v['avg_us'],1)) for n,v in j['kernels'].items() if 'sliced' in n})"
  done
done
