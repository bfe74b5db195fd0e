#!/bin/bash
# Per-kernel register / LDS / occupancy table of one source file (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: tools/resusage.sh emernerf_amd/csrc/mlp_fused.hip [grep pattern] [extra flags]
SRC=$1; PAT=${2:-.}; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function "$@" -c $SRC -o /tmp/_res.o -Rpass-analysis=kernel-resource-usage 2> /tmp/_res.log
grep -E "error" /tmp/_res.log | head
grep -E "Function Name|VGPRs:|AGPRs|Spill|Occupancy|LDS Size" /tmp/_res.log | sed 's/.*remark: [^ ]* *//' | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | paste - - - - - - - | awk '{$1=$1; print}' | sed 's/Function Name: //; s/_ZN4emer//; s/AGPRs: 0 //; s/SGPRs Spill: 0 //; s/LDS Size .*//' | grep -E "$PAT" | cut -c1-160
