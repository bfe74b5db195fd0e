cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_a_metric_shape_gpu.py tests/test_kernels_gpu.py -x -q -k "hashgrid" 2>&1 | tail -4 ) > gpurun_out/t3.log 2>&1
cat gpurun_out/t3.log
( for g in 3,16,16,2048,19,2 3,10,16,8192,20,4 4,10,32,8192,18,4; do for t in base nocarry; do if [ $t = base ]; then L=""; else L="--lib $t"; fi; timeout 200 python tools/grid_only.py --iters 8 --grid $g $L 2>/dev/null | tail -1; done; done ) > gpurun_out/ab7.log 2>&1
cat gpurun_out/ab7.log
