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
ls -la tools/libsglk_abl*.so 2>/dev/null || true

# ATTN=1 tools/build_abl.sh  also builds tools/libsglk_attnstamps.so: the product library with the decode-attention
# phase stamps compiled in (tools/bench_attn.py --stamps under SGLK_LIB=tools/libsglk_attnstamps.so)
if [ -n "$ATTN" ]; then
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Iinclude -I$C -DSGLK_ATTN_STAMPS $EXTRA -c $C/attention_decode.hip -o /tmp/attn_stamps.o
  objs=$(ls $C/build/*.o | grep -v "/attention_decode.hip.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libsglk_attnstamps.so $objs /tmp/attn_stamps.o
  ls -la tools/libsglk_attnstamps.so
fi
