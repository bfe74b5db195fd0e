#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05s16; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_golden_gpu.py tests/test_trainer_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
B="--no-cpu-baseline --no-extras --no-second-state --no-secondary --no-fp16-state --steps 60 --warmup 10"
for r in 1 2 3; do
  for v in 1 0; do
    EMER_FUSE_SAMPLE_POINTS=$v timeout 300 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('static fuse_points=$v graph', round(d['ms_per_step'],4), 'eager', round((d['config']['other_launch_mode'] or {}).get('ms_per_step',0),4))" >> $O/ab.txt
  done
done
cat $O/ab.txt
