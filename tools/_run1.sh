cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_fused_gpu.py -x -q -k "seq_mlp" 2>&1 | tail -3 ) > gpurun_out/t4.log 2>&1
cat gpurun_out/t4.log
timeout 300 python bench.py --kind flow --no-cpu-baseline --no-extras --no-second-state --steps 20 --warmup 5 2>gpurun_out/bf_err.log > gpurun_out/bf.json; tail -3 gpurun_out/bf_err.log; python -c "
import json; b=json.loads(open('gpurun_out/bf.json').read().strip().splitlines()[-1]); print(b['value'], b['config']['workload'][:150]); print(b['roofline']['avg_us'], b['roofline']['grid_encode_plus_bwd']); print(b['roofline_xyzt'])"
