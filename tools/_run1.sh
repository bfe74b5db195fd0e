cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_fused_gpu.py tests/test_golden_gpu.py tests/test_trainer_gpu.py -x -q 2>&1 | tail -15 ) > gpurun_out/t2.log 2>&1
cat gpurun_out/t2.log
for k in dynamic flow feature; do timeout 300 python bench.py --kind $k --no-cpu-baseline --no-extras --no-second-state --steps 12 --warmup 4 2>/dev/null | python -c "
import sys, json
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=b['kernels']
print('$k', round(b['ms_per_step'],3), round(b['value']))
for n,v in sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step'])[:9]: print('   ', n, round(v['launches_per_step'],2), round(v['ms_per_step'],3))
"; done > gpurun_out/kinds.log 2>&1
cat gpurun_out/kinds.log
