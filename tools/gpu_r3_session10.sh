#!/bin/bash
# round 3: saveSpz / loadSpz laps (overlap with the kernel copy; loadSpz stage times)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
SPZ_AMD_EXACT_GZIP_TIMING=1 SPZ_AMD_LZ_TIMING=1 timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 3 1 > $O/host_bench_s10.json 2> $O/host_bench_s10.err; echo "host_bench rc=$?"; cat $O/host_bench_s10.json; grep -E "lz77\]|saveSpz\]|exactgz\] (head|writer)|loadSpz\]|inflate\]" $O/host_bench_s10.err | tail -n 40
timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 3 1 > $O/host_bench_s10b.json 2> $O/host_bench_s10b.err; echo "host_bench (quiet) rc=$?"; cat $O/host_bench_s10b.json
timeout -k 10 300 python -m pytest tests/test_gpu_gzip_device.py tests/test_gpu_container_safety.py -x -q > $O/pytest_s10.log 2>&1; echo "pytest rc=$?"; tail -n 5 $O/pytest_s10.log
