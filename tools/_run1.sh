cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 400 python -m pytest tests/test_a_metric_shape_gpu.py tests/test_kernels_gpu.py -x -q -k "hashgrid" 2>&1 | tail -3 ) > gpurun_out/t1.log 2>&1
( bash tools/ab_grid.sh "base prev base prev" 1 ) > gpurun_out/ab3.log 2>&1
cat gpurun_out/t1.log gpurun_out/ab3.log
