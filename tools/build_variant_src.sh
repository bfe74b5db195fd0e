#!/bin/bash
# Build emernerf_amd/lib/libemernerf_<tag>.so with ONE source file replaced by an arbitrary file (e.g. `git show HEAD:...`).
# usage: tools/build_variant_src.sh <tag> <file.hip to replace> <path of the replacement source> [extra hipcc flags]
set -e
TAG=$1; SRC=$2; REPL=$3; shift 3
R=$(cd "$(dirname "$0")/.." && pwd)
TMP=$R/emernerf_amd/csrc/_variant_$TAG.hip
cp $REPL $TMP
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics "$@" -c $TMP -o /tmp/_variant_$TAG.o
rm $TMP
OBJS=""
for o in $R/emernerf_amd/lib/obj/*.o; do
  if [ "$(basename $o)" != "$SRC.o" ]; then OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/emernerf_amd/lib/libemernerf_$TAG.so /tmp/_variant_$TAG.o $OBJS
echo built libemernerf_$TAG.so
