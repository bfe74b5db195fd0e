#!/bin/bash
# [r6] Grid backward: a level's 64 slice owners dealt to ONE XCD (scheduling blocks of 64 items = two rounds of an XCD's 32 owners on the same
# level: x / dout lines of a level fetched into one L2 instead of two) vs the blocks of 32 (base).  Timing + L2 / HBM counters.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for r in 1 2 3; do
  for t in base sb64; do
    if [ "$t" = "base" ]; then L=""; else L="--lib $t"; fi
    timeout 200 python tools/grid_only.py --iters 12 $L 2>/dev/null | tail -1
    timeout 200 python tools/grid_only.py --iters 12 --grid 4,10,32,8192,18,4 $L 2>/dev/null | tail -1
  done
done
cd /tmp && export TMPDIR=/tmp
for t in base sb64; do
  if [ "$t" = "base" ]; then L=""; else L="--lib $t"; fi
  for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf /tmp/pm_$t; timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pm_$t -o p --output-format csv -- python $R/tools/grid_only.py --iters 6 $L > /dev/null 2>&1
    python - /tmp/pm_$t "$t" "$C" <<'PY'
import csv, glob, sys, collections
d, tag, cs = sys.argv[1:4]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: [0.0, 0])
for fn in f:
    for r in csv.DictReader(open(fn)):
        if 'bwd_params_sliced_kernel<3, 2>' in r['Kernel_Name']:
            a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
print(tag, {k: round(v[0] / max(v[1], 1)) for k, v in agg.items()}, '(per launch; FETCH_SIZE / WRITE_SIZE raw units of the guide: see tools/summarize_pmc.py)')
PY
  done
done
