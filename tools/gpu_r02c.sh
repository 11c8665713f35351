#!/bin/bash
# round 2, GPU session C: full parity suite (incl. IPC / RCCL routes), host path with huge-page prefault,
# per-kernel profile rows, then the bench's own rocprof passes + the bench line.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r02c}
mkdir -p $O
cd $R
timeout -k 10 700 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_$TAG.log 2>&1
rc=$?
tail -n 15 $O/pytest_gpu_$TAG.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 120 ./build/pcie_probe > $O/pcie_probe_$TAG.jsonl 2>&1 || { echo "pcie_probe failed"; tail -n 5 $O/pcie_probe_$TAG.jsonl; exit 3; }
grep -E "hugepage|untouched|touched_pages|thp" $O/pcie_probe_$TAG.jsonl
timeout -k 10 200 ./spz_amd/bin/host_bench 10000000 3 4 > $O/host_bench_$TAG.json 2>&1 || { echo "host_bench failed"; tail -n 5 $O/host_bench_$TAG.json; exit 3; }
cat $O/host_bench_$TAG.json
bash tools/gpu_profile_all.sh $TAG || exit 4
SKIP_TUNE=1 SKIP_PYTEST=1 bash tools/gpu_round.sh $TAG
