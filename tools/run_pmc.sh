#!/bin/bash
# usage: bash tools/run_pmc.sh "<kernel substring>" <python script + args...>   -- SQ counters for one kernel
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pm in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/pmc; (cd $R && timeout 200 rocprofv3 --pmc $pm --kernel-trace -d /tmp/pmc -o p --output-format csv -- python "$@" > /dev/null 2>&1)
  python - "$PAT" <<'PY'
import csv,glob,collections,sys
pat=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for fn in glob.glob('/tmp/pmc/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name']
        if pat not in k: continue
        k=k[:60]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k,v in agg.items():
    print(k, {c: round(x/cnt[(k,c)]) for c,x in v.items()})
PY
done
