#!/bin/bash
# Build emernerf_amd/lib/libemernerf_<tag>.so from a sed-patched copy of ONE source file (for tools/ab_bench.py).
# usage: tools/build_variant.sh <tag> <file.hip> '<sed expression>'
set -e
TAG=$1; SRC=$2; EXPR=$3
R=$(cd "$(dirname "$0")/.." && pwd)
TMP=$R/emernerf_amd/csrc/_variant_$TAG.hip
sed "$EXPR" $R/emernerf_amd/csrc/$SRC > $TMP
if cmp -s $TMP $R/emernerf_amd/csrc/$SRC; then echo "sed expression changed nothing"; rm $TMP; exit 1; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -c $TMP -o /tmp/_variant_$TAG.o
rm $TMP
OBJS=""
for o in $R/emernerf_amd/lib/obj/*.o; do
  if [ "$(basename $o)" != "$SRC.o" ]; then OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/emernerf_amd/lib/libemernerf_$TAG.so /tmp/_variant_$TAG.o $OBJS
echo built libemernerf_$TAG.so
