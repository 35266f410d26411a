#!/bin/bash
set -e
cd "$(dirname "$0")/.."
timeout -k 10 400 python -u -m pytest tests/test_gpu_attention_extend.py tests/test_gpu_backend.py tests/test_gpu_fusion.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
for d in 0 100; do
  echo "== delay $d"
  SGLK_TAIL_DELAY=$d timeout -k 10 120 python tools/bench_attn.py --splits 8 --planned
  SGLK_TAIL_DELAY=$d SGLK_LIB=tools/libsglk_attnstamps.so timeout -k 10 120 python tools/bench_attn.py --splits 8 --stamps --planned | grep -E "end|stream|wait|walk|merge"
done
timeout -k 10 120 python tools/bench_attn.py --splits 8
for c in 0 0; do
  timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-prefix-hit --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['per_projection_us'])"
done
