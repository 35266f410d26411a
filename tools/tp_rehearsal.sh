#!/bin/bash
# Tensor-parallel rehearsal on ONE GPU: N processes (default 8) share the card, the device group is gloo (RCCL refuses
# two ranks on one device), the row-parallel sums and the logits gather run on the peer-to-peer kernels.
# The times say nothing about xGMI (the ranks time-slice one chip); the line shows what ran: ranks, collective, calls.
set -e -o pipefail
N=${1:-8}
TAG=${2:-tp_rehearsal}
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 ${TP_TIMEOUT:-700} python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
  --master-port 29561 bench.py --gpus $N --model llama3-70b-tp8-rank --quant fp8 --tp --dist-backend gloo \
  --steps ${STEPS:-8} --warmup 3 > gpurun_out/${TAG}_n$N.json 2> gpurun_out/${TAG}_n$N.err
tail -c 3000 gpurun_out/${TAG}_n$N.json
