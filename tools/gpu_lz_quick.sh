#!/bin/bash
# quick look at the LZ stage after a kernel change: exactness tests, then one saveSpz with the tables and matches NOT
# beside the upload (their own durations) and the default one
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_gzip_device.py -x -q 2>&1 | tail -n 2 || exit 1
SPZ_AMD_GZIP_OVERLAP=0 SPZ_AMD_LZ_TIMING=1 SPZ_AMD_EXACT_GZIP_TIMING=1 timeout -k 10 200 ./spz_amd/bin/host_bench 10000000 3 2 1 2>&1 >/dev/null | grep -E "walks|tables\+matches|in all \(" | tail -n 3
SPZ_AMD_LZ_TIMING=1 SPZ_AMD_EXACT_GZIP_TIMING=1 timeout -k 10 200 ./spz_amd/bin/host_bench 10000000 3 3 1 2> $O/hb_quick.err | python3 -c "import json,sys; h=json.load(sys.stdin); print('save first/steady, load:', h['save_spz_first_s'], h['save_spz_s'], h['load_spz_s'])"
grep -E "saveSpz\] pack|tables\+matches|writer|in all \(" $O/hb_quick.err | tail -n 4
