SGLK_LIB=tools/libsglk_attnnew.so timeout -k 10 300 python -u -m pytest tests/test_gpu_attention_extend.py tests/test_gpu_fusion.py -m gpu -x -q --timeout 120 --timeout-method thread -k "shared or fus" > gpurun_out/attn3_tests.log 2>&1 || exit 1
echo > gpurun_out/attn3.log
for v in old new old new old new; do
  echo "== $v" >> gpurun_out/attn3.log
  lib=sglang_security_amd/libsglk.so; [ $v = new ] && lib=tools/libsglk_attnnew.so
  SGLK_LIB=$lib timeout -k 10 120 python tools/bench_attn.py >> gpurun_out/attn3.log 2>&1 || exit 1
done
