#!/bin/bash
# Same-session A/B of this round's switchable changes on ONE box (box-to-box variance ~5 %): the rgb head's layer-0/1 weight gradients
# (streamed passes of round 3 / in-kernel, two paired tiles per step / in-kernel, one tile per step = default) on the static step, and the
# grid input gradient (gather pass / stored Jacobians = default) on the flow step at the 2048-ray shard.
# usage (through gpurun): bash tools/ab_r04.sh [rounds] > gpurun_out/r04_ab_step.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; ROUNDS=${1:-2}
B="--no-extras --no-secondary --no-cpu-baseline --no-second-state --no-fp16-state --steps 60 --warmup 10"
show() {
python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']
print('$1', 'graph', round(j['ms_per_step'],4), 'eager', round((j['config']['other_launch_mode'] or {}).get('ms_per_step',0),4), {n.replace('emer_',''): round(v['ms_per_step'],4) for n,v in k.items() if v['ms_per_step']>0.1})"
}
for r in $(seq 1 $ROUNDS); do
  EMER_FUSE_RGB_WGRAD=0 timeout 200 python $R/bench.py $B 2>/dev/null | show "static rgb-wgrad=streamed(r3)"
  EMER_RGBW_PAIR=1 timeout 200 python $R/bench.py $B 2>/dev/null | show "static rgb-wgrad=paired-tiles  "
  timeout 200 python $R/bench.py $B 2>/dev/null | show "static rgb-wgrad=tile(default) "
  EMER_GRID_JAC=0 timeout 200 python $R/bench.py --kind flow --rays 2048 $B 2>/dev/null | show "flow@2048 dx=gather-pass(r3)   "
  timeout 200 python $R/bench.py --kind flow --rays 2048 $B 2>/dev/null | show "flow@2048 dx=jacobians(default)"
done
