#!/bin/bash
# Per-kernel statistics AND the ordered kernel sequence of one BASELINE config at a given shard, from ONE traced run of the eager step
# (rocprofv3 --kernel-trace --stats).  Usage (through gpurun): bash tools/profile_config.sh <tag> <kind> <rays>
#   -> gpurun_out/prof_<tag>/<tag>_<kind><rays>_kernel_stats.csv, ..._step_sequence.txt   (copy into profiles/)
TAG=${1:-r04}; KIND=${2:-flow}; RAYS=${3:-2048}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
D=/tmp/pc_${KIND}_${RAYS}; rm -rf $D
STEPS=12
timeout 400 rocprofv3 --kernel-trace --stats -d $D -o p --output-format csv -- python $R/bench.py --kind $KIND --rays $RAYS --eager --no-cpu-baseline \
  --no-extras --no-second-state --no-secondary --no-fp16-state --steps $STEPS --warmup 2 --init-steps 14 > $OUT/${KIND}${RAYS}_bench_under_rocprof.log 2>&1
cp $(find $D -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_${KIND}${RAYS}_kernel_stats.csv 2>/dev/null
python - $D > $OUT/${TAG}_${KIND}${RAYS}_step_sequence.txt <<'P'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'adam_kernel' in n]
# a step ends with the adam launch of the main group: the last launch before a gap in adam launches -> take the last 3 steps by the
# zero-fill that starts every step (multi_tensor zero) if present, else by adam
cut = idx[-4:]
for a, b in zip(cut[:-1], cut[1:]):
    span = rows[a + 1:b + 1]
    t = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in span) / 1e6
    wall = (int(span[-1]['End_Timestamp']) - int(span[0]['Start_Timestamp'])) / 1e6
    print(f'==== step: {len(span)} launches, {t:.3f} ms of kernels in {wall:.3f} ms')
    for r in span:
        n = r['Kernel_Name'].replace('void ', '').replace('at::native::', '')[:110]
        print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000:8.1f}  {n}")
last = rows[cut[-2] + 1:cut[-1] + 1]
agg = collections.Counter(); cnt = collections.Counter()
for r in last:
    n = r['Kernel_Name'].replace('void ', '').split('(')[0][:70]
    agg[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; cnt[n] += 1
print('==== last step by kernel (us, launches)')
for n, v in agg.most_common(40):
    print(f'{v:9.1f} {cnt[n]:4d}  {n}')
P
python $R/tools/show_stats.py $OUT/${TAG}_${KIND}${RAYS}_kernel_stats.csv $((STEPS + 16 + 12)) 28
tail -45 $OUT/${TAG}_${KIND}${RAYS}_step_sequence.txt
