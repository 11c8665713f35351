#!/bin/bash
# round 3: placement "spread" experiment, then the device inflate tests on the new window / decode kernels with stage laps
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
KINDS=spread64,separate bash tools/gpu_placement_scan.sh g > $O/scan_g.log 2>&1; tail -n 4 $O/scan_g.log
timeout -k 10 600 python -m pytest tests/test_gpu_inflate_device.py tests/test_gpu_container_safety.py -x -q > $O/pytest_inflate.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 15 $O/pytest_inflate.log
[ $rc -eq 0 ] || exit $rc
SPZ_AMD_LZ_TIMING=1 timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 3 1 > $O/host_bench_s2.json 2> $O/host_bench_s2.err; echo "host_bench rc=$?"; cat $O/host_bench_s2.json; grep -E "inflate\]|lz77\]" $O/host_bench_s2.err | tail -n 60
