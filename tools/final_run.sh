#!/bin/bash
# Round-end validation on the GPU box: tools/final_run.sh <tag>  (every GPU step under its own time limit, stop at the first failure)
tag=$1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > gpurun_out/${tag}_tests.log 2>&1 || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1 || exit 2
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench.log 2>&1 || exit 3
bash tools/prof_bench.sh ${tag} > gpurun_out/${tag}_prof.txt 2>&1 || exit 4
bash tools/workloads_run.sh ${tag} || exit 5
