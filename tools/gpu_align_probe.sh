#!/bin/bash
# decode time against buffer placement: bash tools/gpu_align_probe.sh [trials]
cd ${GRAFT_REPO_ROOT:-.} && timeout -k 10 500 python tools/placement_probe.py ${1:-48} > gpurun_out/placement_probe.jsonl 2>&1; tail -n 5 gpurun_out/placement_probe.jsonl | cut -c1-200
