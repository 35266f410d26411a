#!/bin/bash
# A/B of one environment switch inside one GPU call: tools/ab_env.sh VAR A B [bench args]; alternates VAR=A and VAR=B,
# three rounds, prints ms per step.
var=$1; a=$2; b=$3; shift 3
for r in 1 2 3; do
  for v in "$a" "$b"; do
    env "$var=$v" timeout -k 10 200 python bench.py --no-cpu-baseline --no-prefix-hit --steps 40 --warmup 8 "$@" > /tmp/ab.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
print("$var=$v", d["ms_per_step"], d["value"])
PY
  done
done
