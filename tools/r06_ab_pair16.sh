#!/bin/bash
# [r6] Same-session A/B: x-pair gathers of 16-byte (F = 4, fp32) entries on the hashed power-of-two levels of the forward (base) vs two
# independent gathers (libemernerf_nopair16.so = -DEMER_PAIR16=0).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for r in 1 2; do
  for t in base nopair16; do
    for grid in "3,10,16,8192,20,4" "4,10,32,8192,18,4" "4,10,16,4096,18,4"; do
      EMER_LIBSEL_SAME_ABI=1 timeout 200 python tools/r06_fwd_probe.py --lib $t --grid $grid 2>/dev/null | tail -1
    done
  done
done
for r in 1 2 3; do
  for t in base nopair16; do
    EMER_LIBSEL_SAME_ABI=1 timeout 400 python tools/ab_bench.py $t --kind dynamic --no-extras --no-second-state --no-secondary --no-fp16-state --steps 24 --warmup 6 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']
print('dynamic@8192 $t', 'ms/step', round(j['ms_per_step'],4), 'median', round(j['ms_per_step_median'],4), 'grid fwd ms/step', round(k['emer_hashgrid_fwd']['ms_per_step'],4), 'x', k['emer_hashgrid_fwd']['launches_per_step'])"
  done
done
for r in 1 2 3; do
  for t in base nopair16; do
    EMER_LIBSEL_SAME_ABI=1 timeout 400 python tools/ab_bench.py $t --kind flow --rays 2048 --no-extras --no-second-state --no-secondary --no-fp16-state --steps 24 --warmup 6 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']
print('flow@2048 $t', 'ms/step', round(j['ms_per_step'],4), 'median', round(j['ms_per_step_median'],4), 'fwd_jac avg us', round(k.get('emer_hashgrid_fwd_jac',{}).get('avg_us',0),1), 'fwd ms/step', round(k['emer_hashgrid_fwd']['ms_per_step'],4))"
  done
done
