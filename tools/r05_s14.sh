#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05s14; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_a_metric_shape_gpu.py -x -q -k "neck or heads or full_step" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
B="--no-cpu-baseline --no-extras --no-second-state --no-secondary --no-fp16-state --steps 24 --warmup 4"
for r in 1 2 3; do
  timeout 300 python bench.py --kind feature --rays 2048 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('feature2048', round(d['ms_per_step'],3), round((d['config']['other_launch_mode'] or {}).get('ms_per_step',0),3))" >> $O/ab.txt
done
cat $O/ab.txt
