echo > gpurun_out/abl7.log
for a in base 16384 base 16384 base 16384; do
  echo "== $a" >> gpurun_out/abl7.log
  lib=tools/libsglk_abl$a.so; [ $a = base ] && lib=sglang_security_amd/libsglk.so
  SGLK_LIB=$lib timeout -k 10 120 python tools/bench_gemm.py --fused --iters 40 >> gpurun_out/abl7.log 2>&1 || exit 1
done
