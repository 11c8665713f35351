#!/bin/bash
# the device LZ77 parse on the GPU box: its parity tests, then the 650 MB bench with stage laps
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-lz1}
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_gzip_device.py -q -x > $O/pytest_lz_$TAG.log 2>&1
rc=$?
tail -n 15 $O/pytest_lz_$TAG.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 600 python tools/gzip_device_bench.py ${2:-10000000} 3 > $O/gzip_device_$TAG.json 2> $O/gzip_device_$TAG.err || { echo "bench failed"; tail -n 20 $O/gzip_device_$TAG.err; exit 3; }
cat $O/gzip_device_$TAG.json
grep -E "lz77|exactgz|bench" $O/gzip_device_$TAG.err | tail -60
SPZ_AMD_LZ_TIMING=1 SPZ_AMD_EXACT_GZIP_TIMING=1 timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 3 1 > $O/host_bench_$TAG.json 2> $O/host_bench_$TAG.err || { echo "host_bench failed"; tail -n 5 $O/host_bench_$TAG.err; exit 3; }
cat $O/host_bench_$TAG.json
grep -E "lz77|exactgz" $O/host_bench_$TAG.err | tail -20
timeout -k 10 900 python tools/gzip_campaign.py --inputs ${CAMPAIGN_INPUTS:-400} --seed 5 --writer device > $O/gzip_campaign_device_$TAG.json 2> $O/gzip_campaign_device_$TAG.err || { echo "device campaign failed"; tail -n 5 $O/gzip_campaign_device_$TAG.json; exit 4; }
cat $O/gzip_campaign_device_$TAG.json
