cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pm in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  rm -rf /tmp/pmc; timeout 120 rocprofv3 --pmc $pm --kernel-trace -d /tmp/pmc -o p --output-format csv -- python $R/tools/probe_bwd.py 2048 2 > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/pmc/**/*counter_collection.csv',recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'][:50]
        if 'sliced' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k,v in agg.items():
    print(k, {c: round(x/cnt[(k,c)]) for c,x in v.items()})
PY
done
