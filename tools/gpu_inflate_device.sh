#!/bin/bash
# the device inflate on the GPU box: its tests, then loadSpz of the 10 M cloud with laps
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-inf1}
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_inflate_device.py -q -x > $O/pytest_inf_$TAG.log 2>&1
rc=$?
tail -n 25 $O/pytest_inf_$TAG.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
SPZ_AMD_LZ_TIMING=1 SPZ_AMD_PINFLATE_TIMING=1 timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 3 1 > $O/host_bench_$TAG.json 2> $O/host_bench_$TAG.err || { echo "host_bench failed"; tail -n 5 $O/host_bench_$TAG.err; exit 3; }
cat $O/host_bench_$TAG.json
grep -E "inflate" $O/host_bench_$TAG.err | tail -12
SPZ_AMD_LZ_TIMING=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> $O/bench_wf_$TAG.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['whole_file'])"
grep -E "inflate\]" $O/bench_wf_$TAG.err | tail -9
