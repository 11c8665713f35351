#!/bin/bash
# round 3: device inflate tests on the tight-chain wave decoder, then laps with and without the match copies
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_inflate_device.py tests/test_gpu_container_safety.py -x -q > $O/pytest_inflate4.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 8 $O/pytest_inflate4.log
[ $rc -eq 0 ] || exit $rc
SPZ_AMD_LZ_TIMING=1 timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 2 1 > $O/host_bench_s4.json 2> $O/host_bench_s4.err; echo "host_bench rc=$?"; cat $O/host_bench_s4.json; grep -E "inflate\]" $O/host_bench_s4.err | tail -n 9
SPZ_AMD_INFLATE_EXPERIMENT=1 SPZ_AMD_LZ_TIMING=1 timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 2 1 > $O/host_bench_s4x.json 2> $O/host_bench_s4x.err; echo "host_bench (no copies) rc=$?"; grep -E "inflate\]" $O/host_bench_s4x.err | tail -n 9
