#!/bin/bash
# A/B of the shared-prefix kernel forms inside one GPU call: the step with the plan derived on the device, with the
# plan laid out from the runner's groups, and with the kernel reading the group arrays itself.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for mode in derived legacy given derived legacy; do
  echo "== SGLK_SHARED_PLAN=$mode"
  SGLK_SHARED_PLAN=$mode timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-prefix-hit --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['per_projection_us'])"
done
