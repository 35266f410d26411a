#!/bin/bash
# usage: ar_stress.sh <ranks> <tag>   (extra environment is passed through: SGLK_AR_GENERIC, CALLS, ITERS, TMO)
N=${1:-8}; TAG=${2:-ar_stress}
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 WORLD_SIZE=$N
pids=()
for r in $(seq 0 $((N-1))); do
  RANK=$r timeout -k 10 ${LIMIT:-200} python -u tools/ar_stress.py > gpurun_out/${TAG}_r$r.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
for r in $(seq 0 $((N-1))); do tail -n 1 gpurun_out/${TAG}_r$r.log; done
exit $rc
