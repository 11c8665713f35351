#!/bin/bash
# Per-kernel rocprofv3 rows for every kernel of libspz_amd.so:  bash tools/gpu_profile_all.sh <tag>
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pass in stats fetch write sq; do
  case $pass in
    stats) ARGS="--kernel-trace --stats";;
    fetch) ARGS="--pmc FETCH_SIZE --kernel-trace";;
    write) ARGS="--pmc WRITE_SIZE --kernel-trace";;
    sq) ARGS="--pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace";;
  esac
  timeout -k 10 280 rocprofv3 $ARGS --output-format csv -d $O/prof_all_${pass}_$TAG -- python3 $R/tools/all_kernels.py --manifest $O/all_kernels_manifest_${TAG}_$pass.json > $O/all_kernels_${pass}_$TAG.log 2>&1 || { echo "rocprof $pass failed"; tail -n 8 $O/all_kernels_${pass}_$TAG.log; exit 4; }
done
cp $O/all_kernels_manifest_${TAG}_stats.json $O/all_kernels_manifest_$TAG.json
cd $R && python3 tools/summarize_all_kernels.py $TAG
# the same SQ counters for the build with plain IEEE divisions in the quaternion code (tools/tune.py variant), SH0 only:
# the before/after of the quaternion arithmetic in instructions per wave
if [ -f $R/build/variants/libspz_amd_quat_ieee.so ]; then
  cd /tmp
  SPZ_AMD_LIB=$R/build/variants/libspz_amd_quat_ieee.so timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/prof_all_sq_${TAG}ieee -- python3 $R/tools/all_kernels.py --cases "10M sh0,cfg2" --manifest $O/all_kernels_manifest_${TAG}ieee.json > $O/all_kernels_sq_${TAG}ieee.log 2>&1 || echo "IEEE-variant SQ pass failed (ignored)"
  cd $R && python3 tools/summarize_all_kernels.py ${TAG}ieee || true
fi
