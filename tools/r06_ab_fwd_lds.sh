#!/bin/bash
# [r6] Same-session A/B: leading dense levels of the grid forward encoded from an LDS-staged copy (base) vs every level through the texture
# path (libemernerf_nolds.so = -DEMER_FWD_LDS_STAGE=0).  Kernel-level (tools/grid_only.py) and the full static step.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for r in 1 2; do
  for t in base nolds; do
    for grid in "3,8,16,512,20,1" "3,8,16,2048,20,1" "3,16,16,2048,19,2" "3,10,16,8192,20,4"; do
      EMER_LIBSEL_SAME_ABI=1 timeout 200 python tools/r06_fwd_probe.py --lib $t --grid $grid 2>/dev/null | tail -1
    done
  done
done
for r in 1 2 3; do
  for t in base nolds; do
    EMER_LIBSEL_SAME_ABI=1 timeout 300 python tools/ab_bench.py $t --no-extras --no-second-state --no-secondary --no-fp16-state --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); rp=j['roofline_prop']
print('step $t', 'ms/step', round(j['ms_per_step'],4), 'median', round(j['ms_per_step_median'],4), 'prop fwd ms/step', round(rp['ms_per_step']['fwd'],4), 'frac_fwd', round(rp['frac_fwd'],3), 'main fwd us', round(j['roofline']['grid_encode_plus_bwd']['fwd_avg_us'],1))"
  done
done
