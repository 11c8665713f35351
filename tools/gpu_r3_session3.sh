#!/bin/bash
# round 3: tile-order variants in a slow and in fast placements; device inflate with and without its match copies
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
VARIANTS=seq,il_g8,il_g64,il_g256,il_g512,il_rot_g8,il_rot_g64 KINDS=variants bash tools/gpu_placement_scan.sh h > $O/scan_h.log 2>&1; tail -n 12 $O/scan_h.log
SPZ_AMD_LZ_TIMING=1 timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 2 1 > $O/host_bench_s3.json 2> $O/host_bench_s3.err; echo "host_bench rc=$?"; grep -E "inflate\]" $O/host_bench_s3.err | tail -n 9
SPZ_AMD_INFLATE_EXPERIMENT=1 SPZ_AMD_LZ_TIMING=1 timeout -k 10 300 ./spz_amd/bin/host_bench 10000000 3 2 1 > $O/host_bench_s3x.json 2> $O/host_bench_s3x.err; echo "host_bench (no copies) rc=$?"; grep -E "inflate\]" $O/host_bench_s3x.err | tail -n 9
