#!/bin/bash
# MFMA-utilisation counters and the instruction mix of the head kernels (field / neck / rgb / plain heads / weight gradients) over a
# short eager bench run.
# Counters in their own pass with --kernel-trace only.  Usage (gpurun): bash tools/pmc_heads.sh <tag> [bench args]
#   -> gpurun_out/pmc_heads_<tag>/summary.json  (copy to profiles/<tag>_mfma_counters.json)
TAG=${1:-r03}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_heads_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmch /tmp/pmch2
BARGS="--eager --no-cpu-baseline --no-extras --no-second-state --no-secondary --no-fp16-state --steps 12 --warmup 4 --init-steps 12"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmch -o p --output-format csv -- \
  python $R/bench.py $BARGS "$@" > $OUT/pass.log 2>&1
echo "rc=$?" >> $OUT/pass.log
# second pass (own run): instruction mix -- how many vector / matrix / memory instructions a wave issues per launch
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA --kernel-trace -d /tmp/pmch2 -o p --output-format csv -- \
  python $R/bench.py $BARGS "$@" > $OUT/pass2.log 2>&1
echo "rc=$?" >> $OUT/pass2.log
python - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
dur = collections.defaultdict(float); nd = collections.Counter()
for fn in glob.glob('/tmp/pmch/**/*counter_collection.csv', recursive=True) + glob.glob('/tmp/pmch2/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].replace('void ', '').replace('emer::', '')
        if not any(t in k for t in ('neck_', 'rgb_', 'rmlp_', 'wgrad_stream', 'mlp_chain', 'linear_fwd', 'field_fwd', 'density_', 'ray_wgrad')): continue
        k = k.split('(')[0][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for fn in glob.glob('/tmp/pmch/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].replace('void ', '').replace('emer::', '').split('(')[0][:60]
        if k in agg:
            dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3; nd[k] += 1
res = {}
for k, d in agg.items():
    c = {n: v / cnt[(k, n)] for n, v in d.items()}
    us = dur[k] / max(nd[k], 1)
    clock_ghz = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0 / (us * 1e3) if us else 0.0   # GRBM_GUI_ACTIVE sums the 8 XCDs
    busy = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    c.update({"launches": nd[k], "avg_us_traced": us, "effective_clock_ghz": clock_ghz,
              "mfma_busy_frac": busy / (1024.0 * us * 1e3 * clock_ghz) if us and clock_ghz else None})
    res[k] = c
json.dump({"note": "per launch averages; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles at the effective clock "
                   "GRBM_GUI_ACTIVE / 8 / duration); SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 = fp32 matrix flops", "kernels": res},
          open(out + '/summary.json', 'w'), indent=1)
for k, c in sorted(res.items(), key=lambda kv: -kv[1]['avg_us_traced'] * kv[1]['launches'])[:12]:
    print(f"{k:55s} n={c['launches']:4d} {c['avg_us_traced']:8.1f} us  clk {c['effective_clock_ghz']:.2f} GHz  mfma busy {c['mfma_busy_frac'] if c['mfma_busy_frac'] is None else round(c['mfma_busy_frac'], 3)}  mops {c.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0):.3g}")
PY
