#!/bin/bash
# placement experiments of round 3: tools/placement_scan.hip over four kinds of device memory, then a bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
TAG=${1:-a}
timeout -k 10 420 ./build/placement_scan ${KINDS:-malloc,vmm1,vmmN,vmmS} 5 ${SEED:-1} > $O/placement_scan_$TAG.jsonl 2> $O/placement_scan_$TAG.err || { echo "scan failed"; tail -n 5 $O/placement_scan_$TAG.err; exit 3; }
wc -l $O/placement_scan_$TAG.jsonl
python3 tools/placement_scan_summary.py $O/placement_scan_$TAG.jsonl
