#!/bin/bash
# round 3: pick experiment; container + packed-device tests; inflate laps with and without the match copies
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
KINDS=pick bash tools/gpu_placement_scan.sh i > $O/scan_i.log 2>&1; grep "^pick" $O/scan_i.log
timeout -k 10 600 python -m pytest tests/test_gpu_inflate_device.py tests/test_gpu_container_safety.py tests/test_gpu_packed_device.py -q > $O/pytest_s5.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 12 $O/pytest_s5.log
SPZ_AMD_LZ_TIMING=1 timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 2 1 > $O/host_bench_s5.json 2> $O/host_bench_s5.err; echo "host_bench rc=$?"; cat $O/host_bench_s5.json; grep -E "inflate\]" $O/host_bench_s5.err | tail -n 9
SPZ_AMD_INFLATE_EXPERIMENT=1 SPZ_AMD_LZ_TIMING=1 timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 2 1 > $O/host_bench_s5x.json 2> $O/host_bench_s5x.err; echo "host_bench (no copies) rc=$?"; grep -E "inflate\]" $O/host_bench_s5x.err | tail -n 9
