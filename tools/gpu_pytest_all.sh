#!/bin/bash
# the whole -m gpu suite, as the driver runs it at round end
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_${1:-all}.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 15 $O/pytest_gpu_${1:-all}.log
