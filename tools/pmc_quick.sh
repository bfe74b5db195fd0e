#!/bin/bash
# usage: bash tools/pmc_quick.sh "<kernel substring>" "<counters of ONE pass>" <python script + args...>  -- per-launch averages
PAT=$1; PM=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmcq; (cd $R && timeout 200 rocprofv3 --pmc $PM --kernel-trace -d /tmp/pmcq -o p --output-format csv -- python "$@" > /dev/null 2>&1)
python - "$PAT" <<'PY'
import csv,glob,collections,sys
pat=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for fn in glob.glob('/tmp/pmcq/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name']
        if pat not in k: continue
        k=k[:50]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k,v in agg.items():
    print(k, {c: round(x/cnt[(k,c)]) for c,x in v.items()})
PY
