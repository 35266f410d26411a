#!/bin/bash
# The other workloads of bench.py on one GPU, one JSON line each into gpurun_out/<tag>_workloads.log
tag=$1
out=gpurun_out/${tag}_workloads.log
: > $out
B="--no-cpu-baseline --no-prefix-hit --steps 30 --warmup 6"
for args in "--batch 1" "--batch 8" "--batch 16" "--batch 32" "--batch 128" "--quant bf16" "--quant fp8" \
            "--model llama3-70b-tp8-rank --quant fp8" "--model qwen2-0.5b --quant bf16 --batch 1 --prefix 0 --suffix 128" \
            "--model qwen2-0.5b --quant bf16" "--model deepseek-v3-dense" "--attn per_request" "--no-fusion"; do
  echo "## $args" >> $out
  timeout -k 10 200 python bench.py $B $args >> $out 2>> gpurun_out/${tag}_workloads.err || exit 1
done
