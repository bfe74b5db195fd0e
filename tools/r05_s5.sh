#!/bin/bash
# round-5 session 5: fused neck backward with one wave per SIMD for 33-64-feature encodings (default) vs two (variant neck2w)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05s5; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_fused_gpu.py -x -q -k "neck" > $O/pytest_neck.log 2>&1; echo "pytest rc $?" >> $O/pytest_neck.log
tail -4 $O/pytest_neck.log
show() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']
print('$1', round(j['ms_per_step'],4), {n.replace('emer_',''): round(v['ms_per_step'],4) for n,v in k.items() if v['ms_per_step']>0.15})"; }
for r in 1 2; do
  for t in base neck2w; do
    EMER_LIBSEL_SAME_ABI=1 timeout 300 python tools/ab_bench.py $t --kind flow --rays 2048 --no-extras --no-second-state --no-secondary --no-fp16-state --steps 16 --warmup 4 2>/dev/null | show "flow2048 $t" >> $O/ab_neck.txt
    EMER_LIBSEL_SAME_ABI=1 timeout 300 python tools/ab_bench.py $t --kind dynamic --no-extras --no-second-state --no-secondary --no-fp16-state --steps 16 --warmup 4 2>/dev/null | show "dynamic $t" >> $O/ab_neck.txt
  done
done
cat $O/ab_neck.txt
