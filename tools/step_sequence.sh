#!/bin/bash
# Ordered kernel sequence of ONE eager static step (rocprofv3 --kernel-trace), to see which small launches are left.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/seq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/seq
timeout 300 rocprofv3 --kernel-trace -d /tmp/seq -o s --output-format csv -- python $R/bench.py ${KIND:+--kind $KIND} --eager --no-cpu-baseline --no-extras --no-second-state --no-secondary --no-fp16-state --steps 6 --warmup 2 > $OUT/log.txt 2>&1
python - <<'P' > $OUT/sequence.txt
import csv, glob
f = glob.glob('/tmp/seq/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# steps end with the adam kernel of the main group: take the last three full steps
idx = [i for i, n in enumerate(names) if 'adam_kernel' in n]
cut = idx[-4:]
for a, b in zip(cut[:-1], cut[1:]):
    print('==== step', b - a, 'launches')
    for r in rows[a + 1:b + 1]:
        n = r['Kernel_Name']
        n = n.replace('void ', '').replace('at::native::', '')[:110]
        print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000:8.1f}  {n}")
P
tail -120 $OUT/sequence.txt
