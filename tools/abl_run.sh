cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_ttft_prof -o ttft -- python3 tools/bench_ttft.py > gpurun_out/r02_ttft.json 2> gpurun_out/r02_ttft.err
