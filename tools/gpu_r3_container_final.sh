#!/bin/bash
# round 3: the container stage at the final HEAD: its tests, host_bench (laps + quiet), rocprofv3 kernel stats of saveSpz + loadSpz
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r03c}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_gzip_device.py tests/test_gpu_container_safety.py tests/test_gpu_inflate_device.py tests/test_gpu_packed_device.py -x -q > $O/pytest_$TAG.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 4 $O/pytest_$TAG.log
[ $rc -eq 0 ] || exit $rc
SPZ_AMD_LZ_TIMING=1 SPZ_AMD_EXACT_GZIP_TIMING=1 timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 3 1 > $O/host_bench_$TAG.json 2> $O/host_bench_$TAG.err || { echo "host_bench failed"; exit 9; }
cat $O/host_bench_$TAG.json | cut -c1-200
timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 4 1 > $O/host_bench_${TAG}_quiet.json 2>/dev/null; cat $O/host_bench_${TAG}_quiet.json
bash tools/gpu_container_stats.sh $TAG > $O/container_stats_run_$TAG.log 2>&1; tail -n 3 $O/container_stats_run_$TAG.log | cut -c1-200
