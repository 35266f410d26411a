#!/bin/bash
# Hardware-counter passes over the decode step (run on the GPU box): tools/pmc_bench.sh <tag>
# One rocprofv3 --pmc run per counter group (FETCH_SIZE, WRITE_SIZE and the SQ set cannot share a pass); eager launches
# (--no-graph) so that every dispatch is attributed to its kernel.  tools/pmc_traffic.py turns the CSVs into the JSON
# bench.py reads.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/${tag}_pmc_${name} -o ${name} -- \
    python3 bench.py --no-cpu-baseline --no-prefix-hit --no-graph --steps 2 --warmup 2 > gpurun_out/${tag}_pmc_${name}.json 2> gpurun_out/${tag}_pmc_${name}.err || return 1
}
run fetch FETCH_SIZE && run write WRITE_SIZE && \
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES
