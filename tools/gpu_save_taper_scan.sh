#!/bin/bash
# saveSpz 10 M SH3: the size of the sh pieces fed to the container stage beside the upload, and of the last of them
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for cfg in ${SCAN:-"480 480" "480 48" "480 16" "320 48" "240 32" "640 48"}; do
  set -- ${cfg/:/ }
  SPZ_AMD_HOST_SESSION_CHUNK_MIB=$1 SPZ_AMD_HOST_SESSION_LAST_MIB=$2 SPZ_AMD_LZ_TIMING=1 SPZ_AMD_EXACT_GZIP_TIMING=1 timeout -k 10 200 ./spz_amd/bin/host_bench 10000000 3 4 1 2> $O/taper.err | python3 -c "import json,sys; h=json.load(sys.stdin); print('pieces $1 MiB, last $2 MiB: save', h['save_spz_s'])" || exit 1
  grep -E "saveSpz\] pack|tables\+matches" $O/taper.err | tail -n 2 | tr '\n' ' '; echo
done
