#!/bin/bash
set -e
cd "$(dirname "$0")/.."
for c in 0 1 0; do
  echo "== SGLK_CHAIN=$c"
  SGLK_CHAIN=$c timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-prefix-hit --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['per_projection_us'])"
done
