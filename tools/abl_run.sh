cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_ttft_prof3 -o ttft -- python3 tools/ttft_host_profile.py > gpurun_out/r02_ttft3.log 2> gpurun_out/r02_ttft3.err
