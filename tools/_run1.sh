cd $GRAFT_REPO_ROOT
( for t in base abl1 abl8 abl16; do if [ $t = base ]; then L=""; else L="--lib $t"; fi; timeout 200 python tools/grid_only.py --iters 6 --grid 4,10,32,8192,18,4 $L 2>/dev/null | tail -1; done; python tools/kbench.py --help 2>&1 | head -20 ) > gpurun_out/ab6.log 2>&1
cat gpurun_out/ab6.log
