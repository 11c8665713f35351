#!/bin/bash
# loadSpz (C++ boundary, host_bench) against the size of the file: the device reader forced on and off
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for n in ${SIZES:-250000 500000 1000000 2000000 4000000 10000000}; do
  for dev in 1 0; do
    SPZ_AMD_GUNZIP_DEVICE=$dev timeout -k 10 300 ./spz_amd/bin/host_bench $n 3 4 1 2>/dev/null | python3 -c "
import json,sys; h=json.load(sys.stdin); print('points $n  device reader $dev  load', h['load_spz_s'], ' first', h['load_spz_first_s'], ' save', h['save_spz_s'], ' file MB', round(h.get('spz_bytes',0)/1e6,1))" || exit 1
  done
done
