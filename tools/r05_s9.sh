#!/bin/bash
# round-5 session 9: weight-gradient-only launches on an auxiliary stream -- parity of the trainer paths, then the same-session step A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05s9; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_trainer_gpu.py tests/test_fused_gpu.py tests/test_train_parity_gpu.py tests/test_multi_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
B="--no-cpu-baseline --no-extras --no-second-state --no-secondary --no-fp16-state --steps 60 --warmup 10"
for r in 1 2 3; do
  for v in 1 0; do
    EMER_AUX_WGRAD=$v timeout 300 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('static aux=$v graph', round(d['ms_per_step'],4), 'eager', round((d['config']['other_launch_mode'] or {}).get('ms_per_step',0),4))" >> $O/ab_aux.txt
  done
done
for v in 1 0; do
  EMER_AUX_WGRAD=$v timeout 300 python bench.py --kind flow --rays 2048 --no-cpu-baseline --no-extras --no-second-state --no-secondary --no-fp16-state --steps 16 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flow2048 aux=$v', round(d['ms_per_step'],4))" >> $O/ab_aux.txt
done
cat $O/ab_aux.txt
