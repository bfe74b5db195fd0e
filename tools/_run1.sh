cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_a_metric_shape_gpu.py tests/test_kernels_gpu.py -x -q -k "hashgrid" 2>&1 | tail -4 ) > gpurun_out/t3.log 2>&1
cat gpurun_out/t3.log
( for g in 3,10,16,8192,20,4 3,16,16,2048,19,2; do timeout 200 python tools/grid_only.py --iters 8 --grid $g 2>&1 | tail -1; done ) > gpurun_out/grids2.log 2>&1
cat gpurun_out/grids2.log
