#!/bin/bash
# SQ / LDS / TCP counter passes for the two grid kernels (forward + owner-computes backward) on the training sample
# distribution.  Counters in their own runs with --kernel-trace only (never with sys/hip traces).  Usage (gpurun):
#   bash tools/pmc_grid.sh <tag> [grid_only.py args]   -> gpurun_out/pmc_<tag>/summary.json
TAG=${1:-r02}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  rm -rf /tmp/pmcg_$i
  timeout 240 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pmcg_$i -o p --output-format csv -- python $R/tools/grid_only.py "$@" > $OUT/pass_$i.log 2>&1
  echo "pass $i ($SET): rc=$?" >> $OUT/passes.txt
done <<'SETS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN
SQ_INSTS_FLAT SQ_INSTS_GDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum
GRBM_GUI_ACTIVE
SETS
python - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in glob.glob('/tmp/pmcg_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if 'hashgrid' not in k: continue
        k = k.split('(')[0].replace('void ', '').replace('emer::', '')[:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
res = {k: {c: v / cnt[(k, c)] for c, v in d.items()} for k, d in agg.items()}
json.dump({"note": "average per launch; SQ_*_CYCLES / WAIT / ACTIVE count quad-cycles summed over waves (MI355X_MICROARCH.md)", "kernels": res},
          open(out + '/summary.json', 'w'), indent=1)
for k, d in res.items():
    print(k); [print('   ', c, round(v)) for c, v in sorted(d.items())]
PY
cat $OUT/passes.txt
