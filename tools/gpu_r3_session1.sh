#!/bin/bash
# round 3, first functional session: placement chunk experiment, then the container-stage tests on the new code
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
KINDS=chunkbw60,matrix60 bash tools/gpu_placement_scan.sh f > $O/scan_f.log 2>&1; tail -n 3 $O/scan_f.log
timeout -k 10 900 python -m pytest tests/test_gpu_container_safety.py tests/test_gpu_gzip_device.py tests/test_gpu_inflate_device.py -x -q > $O/pytest_safety.log 2>&1
echo "pytest rc=$?"; tail -n 15 $O/pytest_safety.log
