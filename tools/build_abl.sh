#!/bin/bash
# Development aid: build libsglk variants with experiment bits of the wide AWQ GEMM compiled in (SGLK_ABL, see
# awq.hip), next to the product library, for tools/bench_gemm.py under SGLK_LIB=...   usage: tools/build_abl.sh 64 71 ...
# EXTRA="-fno-slp-vectorize" tools/build_abl.sh 0 builds variant 0 with extra compiler options.
set -e
cd "$(dirname "$0")/.."
C=sglang_security_amd/csrc
for a in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Iinclude -I$C -DSGLK_ABL=$a $EXTRA -c $C/awq.hip -o /tmp/awq_abl$a.o &
done
wait
for a in "$@"; do
  objs=$(ls $C/build/*.o | grep -v "/awq.hip.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libsglk_abl$a.so $objs /tmp/awq_abl$a.o
done
ls -la tools/libsglk_abl*.so
