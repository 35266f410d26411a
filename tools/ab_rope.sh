#!/bin/bash
set -e
cd "$(dirname "$0")/.."
for c in "1 100" "1 0" "0 100" "1 50"; do
  set -- $c
  echo "== SGLK_ROPE_ROLE=$1 SGLK_TAIL_DELAY=$2"
  SGLK_ROPE_ROLE=$1 SGLK_TAIL_DELAY=$2 timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-prefix-hit --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"
done
