#!/bin/bash
# rocprofv3 kernel statistics of a short bench run on the GPU box; prints the top kernels.  usage: tools/prof_bench.sh tag [bench args]
tag=$1; shift
/usr/local/graft/bin/gpurun --timeout 900 -- "cd /tmp && export TMPDIR=/tmp && R=\$GRAFT_REPO_ROOT && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d \$R/gpurun_out/$tag -o bench -- python3 \$R/bench.py --no-cpu-baseline --no-prefix-hit --steps 20 --warmup 4 $* > \$R/gpurun_out/$tag.log 2>&1; echo exit=\$?" 2>&1 | grep -E "exit=|GPU-minutes"
python tools/rocpd_stats.py gpurun_out/$tag/bench_results.db > gpurun_out/$tag.csv
python - "$tag" <<'PY'
import csv, sys
rows=list(csv.reader(open(f'gpurun_out/{sys.argv[1]}.csv')))[1:]
for r in rows[:14]:
    if 'spin_kernel' in r[0]: continue
    print(r[0][:58].ljust(58), r[1].rjust(6), f"{float(r[3])/1e3:8.2f} us", f"{float(r[2])/26/1e3:8.1f} us/step")
PY
