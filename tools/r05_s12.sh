#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05s12; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_fused_gpu.py -x -q -k "seq_mlp" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
B="--no-cpu-baseline --no-extras --no-second-state --no-secondary --no-fp16-state --steps 16 --warmup 4"
for r in 1 2; do
  for v in 1 0; do
    EMER_FUSE_RMLP_WGRAD=$v timeout 300 python bench.py --kind feature --rays 2048 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('feature2048 fuse_rmlp=$v', round(d['ms_per_step'],3))" >> $O/ab.txt
  done
done
cat $O/ab.txt
