#!/bin/bash
# final state of the round: the whole GPU suite, smoke, the bench line (with the whole-file figure), host bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r02final}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_$TAG.log 2>&1
rc=$?
tail -n 6 $O/pytest_gpu_$TAG.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 2
timeout -k 10 600 python bench.py > $O/bench_$TAG.log 2> $O/bench_$TAG.err || { echo "bench failed"; tail -n 20 $O/bench_$TAG.err; exit 3; }
tail -n 1 $O/bench_$TAG.log
SPZ_AMD_LZ_TIMING=1 SPZ_AMD_EXACT_GZIP_TIMING=1 timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 3 1 > $O/host_bench_$TAG.json 2> $O/host_bench_$TAG.err || { echo "host_bench failed"; exit 4; }
cat $O/host_bench_$TAG.json
