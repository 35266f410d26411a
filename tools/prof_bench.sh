#!/bin/bash
# Per-kernel statistics of the default bench under rocprofv3 (run on the GPU box): tools/prof_bench.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o ${tag} -- python3 bench.py --no-cpu-baseline --no-prefix-hit --steps 20 --warmup 4 "$@" > gpurun_out/${tag}_under_rocprof.json 2> gpurun_out/${tag}.err
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/${tag}_prof/${tag}_kernel_stats.csv')))
for r in rows[:12]:
    print(f"{r['Name'][:80]:80s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:8.2f}us {r['Percentage']:>6s}%")
PY
