#!/bin/bash
# load path on the GPU box: parity subset, smoke, C++ host bench incl. saveSpz/loadSpz with the inflate laps, gunzip by thread count
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r02j}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_$TAG.log 2>&1
rc=$?
tail -n 6 $O/pytest_gpu_$TAG.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
SPZ_AMD_PINFLATE_TIMING=1 timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 3 1 > $O/host_bench_$TAG.json 2> $O/host_bench_$TAG.err || { echo "host_bench failed"; tail -n 5 $O/host_bench_$TAG.err; exit 3; }
cat $O/host_bench_$TAG.json; grep pinflate $O/host_bench_$TAG.err | tail -8
timeout -k 10 300 python tools/pinflate_bench.py 2000000 > $O/pinflate_$TAG.json 2>/dev/null; cat $O/pinflate_$TAG.json
