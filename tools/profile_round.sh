#!/bin/bash
# Round profile on the GPU box: kernel-trace stats + PMC HBM traffic (separate passes) + the bench line.
# Usage (through gpurun): bash tools/profile_round.sh r01_b
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --no-cpu-baseline --no-extras --no-second-state --no-secondary --no-fp16-state --steps 24 --warmup 8"

rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- $CMD > $OUT/bench_under_rocprof.log 2>&1
cp $(find /tmp/kt -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_bench_kernel_stats.csv 2>/dev/null

for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o p --output-format csv -- $CMD > $OUT/pmc_$C.log 2>&1
  # calibration dispatches with a known byte count in the same pass (see tools/pmc_calibrate.py)
  rm -rf /tmp/cal_$C
  timeout 120 rocprofv3 --pmc $C --kernel-trace -d /tmp/cal_$C -o p --output-format csv -- python $R/tools/pmc_calibrate.py > $OUT/cal_$C.log 2>&1
done
python $R/tools/summarize_pmc.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/cal_FETCH_SIZE /tmp/cal_WRITE_SIZE > $OUT/${TAG}_hbm_traffic.json 2> $OUT/summarize.log
cd $R && timeout 400 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.log
tail -c 600 $OUT/${TAG}_hbm_traffic.json; echo; head -c 400 $OUT/${TAG}_bench.json; echo; head -12 $OUT/${TAG}_bench_kernel_stats.csv | cut -c1-150
