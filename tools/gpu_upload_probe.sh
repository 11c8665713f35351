#!/bin/bash
# several fresh processes: how often is the upload of a fresh vector slow, and what does the adaptive upload make of it
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
for i in 1 2 3; do
  SPZ_AMD_LZ_TIMING=1 timeout -k 10 120 ./spz_amd/bin/host_bench 10000000 3 2 1 > gpurun_out/up_$i.json 2> gpurun_out/up_$i.err
  echo "process $i: $(grep -o '"save_spz_s": [0-9.]*, "load_spz_s": [0-9.]*' gpurun_out/up_$i.json) | $(grep -E '\[upload\]' gpurun_out/up_$i.err | tail -2 | tr '\n' ' ') | $(grep -E 'lz77\] upload|inflate\] upload|\] alloc' gpurun_out/up_$i.err | tail -4 | tr "\n" " ")"
done
