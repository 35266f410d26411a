#!/bin/bash
# A/B of two builds of the library inside one GPU call: tools/ab_lib.sh <other .so> [bench args]; alternates the product
# library and the other one, three rounds, prints ms per step.
other=$1; shift
for r in 1 2 3; do
  for lib in "" "$other"; do
    SGLK_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-prefix-hit --steps 40 --warmup 8 "$@" > /tmp/ab.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
print("${lib:-product}", d["ms_per_step"], d["value"])
PY
  done
done
