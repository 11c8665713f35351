#!/bin/bash
# rocprofv3 kernel statistics of a whole saveSpz + loadSpz (host_bench): the container stage's kernels beside the quantise kernels
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-cs1}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_container_$TAG -o run -- ./spz_amd/bin/host_bench 10000000 3 2 1 > $O/container_stats_$TAG.log 2>&1 || { echo "rocprofv3 failed"; tail -n 8 $O/container_stats_$TAG.log; exit 3; }
f=$(find $O/prof_container_$TAG -name '*kernel_stats.csv' | head -n 1)
[ -n "$f" ] && cp "$f" $O/container_kernel_stats_$TAG.csv && cat $O/container_kernel_stats_$TAG.csv
tail -n 1 $O/container_stats_$TAG.log
