#!/bin/bash
# round-5 session 1: new tests, current step sequences (static / flow at the 2048-ray shard), torch-operator attribution, LUT select A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05s1; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_trainer_gpu.py -x -q > $O/pytest_trainer.log 2>&1; echo "pytest rc $?" >> $O/pytest_trainer.log
tail -3 $O/pytest_trainer.log
bash tools/step_sequence.sh > /dev/null 2>&1; cp gpurun_out/seq/sequence.txt $O/static_sequence.txt
bash tools/profile_config.sh r05a flow 2048 > $O/flow2048.log 2>&1
timeout 200 python tools/torch_ops_probe.py flow 2048 > $O/torch_ops_flow.txt 2>&1
for r in 1 2; do
  for t in base lut; do
    if [ "$t" = "base" ]; then L=""; else L="--lib $t"; fi
    timeout 150 python tools/grid_only.py --iters 12 $L 2>/dev/null | tail -1 >> $O/ab_lut.txt
    timeout 150 python tools/grid_only.py --iters 8 --grid 4,10,32,8192,18,4 $L 2>/dev/null | tail -1 >> $O/ab_lut.txt
  done
done
cat $O/ab_lut.txt
tail -48 $O/static_sequence.txt | cut -c1-120
