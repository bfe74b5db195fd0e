#!/bin/bash
# Same-session A/B of the full static step across library variants of the CURRENT sources: usage: bash tools/ab_step.sh "base tagA tagB" [rounds]
TAGS=${1:-base}; ROUNDS=${2:-2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
for r in $(seq 1 $ROUNDS); do
  for t in $TAGS; do
    EMER_LIBSEL_SAME_ABI=1 timeout 200 python $R/tools/ab_bench.py $t --no-extras --no-second-state --no-secondary --no-fp16-state --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']
print('$t', round(j['ms_per_step'],4), 'eager', round((j['config']['other_launch_mode'] or {}).get('ms_per_step',0),4), {n: round(v['ms_per_step'],4) for n,v in k.items() if v['ms_per_step']>0.2})"
  done
done
