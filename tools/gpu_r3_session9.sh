#!/bin/bash
# round 3: the container tests on the session-fed writer, then saveSpz / loadSpz laps with and without the overlap
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_gzip_device.py tests/test_gpu_container_safety.py tests/test_gpu_inflate_device.py tests/test_gpu_python_module.py -x -q > $O/pytest_s9.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 12 $O/pytest_s9.log
[ $rc -eq 0 ] || exit $rc
SPZ_AMD_EXACT_GZIP_TIMING=1 SPZ_AMD_LZ_TIMING=1 timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 3 1 > $O/host_bench_s9.json 2> $O/host_bench_s9.err; echo "host_bench rc=$?"; cat $O/host_bench_s9.json; grep -E "lz77\]|saveSpz\]|exactgz\]" $O/host_bench_s9.err | tail -n 42
SPZ_AMD_GZIP_OVERLAP=0 timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 3 1 > $O/host_bench_s9n.json 2> $O/host_bench_s9n.err; echo "host_bench (no overlap) rc=$?"; cat $O/host_bench_s9n.json
