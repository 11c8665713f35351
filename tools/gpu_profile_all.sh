#!/bin/bash
# Per-kernel rocprofv3 rows for every kernel of libspz_amd.so:  bash tools/gpu_profile_all.sh <tag>
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pass in stats fetch write; do
  case $pass in
    stats) ARGS="--kernel-trace --stats";;
    fetch) ARGS="--pmc FETCH_SIZE --kernel-trace";;
    write) ARGS="--pmc WRITE_SIZE --kernel-trace";;
  esac
  timeout -k 10 280 rocprofv3 $ARGS --output-format csv -d $O/prof_all_${pass}_$TAG -- python3 $R/tools/all_kernels.py --manifest $O/all_kernels_manifest_${TAG}_$pass.json > $O/all_kernels_${pass}_$TAG.log 2>&1 || { echo "rocprof $pass failed"; tail -n 8 $O/all_kernels_${pass}_$TAG.log; exit 4; }
done
cp $O/all_kernels_manifest_${TAG}_stats.json $O/all_kernels_manifest_$TAG.json
cd $R && python3 tools/summarize_all_kernels.py $TAG
