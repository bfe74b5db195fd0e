#!/bin/bash
# round-5 session 6: full GPU suite on the current sources; flow-table A/B (round-4 kernel vs now); flow / dynamic / static step times
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05s6; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_all.log
tail -6 $O/pytest_all.log
for r in 1 2; do
  for t in r4 base; do
    if [ "$t" = "base" ]; then L=""; else L="--lib $t"; fi
    for G in 4,10,16,4096,18,4 4,10,32,8192,18,4; do
      timeout 150 python tools/grid_only.py --iters 8 --grid $G $L 2>/dev/null | tail -1 | cut -c1-190 >> $O/ab_flowgrid.txt
    done
  done
done
cat $O/ab_flowgrid.txt
B="--no-cpu-baseline --no-extras --no-second-state --no-secondary --no-fp16-state --steps 16 --warmup 4"
for k in "flow --rays 2048" "feature --rays 2048" "dynamic" "static"; do
  timeout 300 python bench.py --kind $k $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$k', round(d['ms_per_step'],3), (d.get('roofline_xyzt') or {}).get('frac_bwd'), d['roofline']['frac'])" >> $O/steps.txt
done
cat $O/steps.txt
timeout 200 python tools/torch_ops_probe.py flow 2048 2>/dev/null | tail -30 > $O/torch_ops_flow.txt
