#!/bin/bash
# saveSpz of 10 M SH3 points against the size of the session path's sh pieces
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
for mib in 160 320 480 640 960; do
  SPZ_AMD_HOST_SESSION_CHUNK_MIB=$mib timeout -k 10 200 ./spz_amd/bin/host_bench 10000000 3 4 1 > $O/chunk_$mib.json 2>/dev/null
  python3 -c "
import json; h=json.load(open('$O/chunk_$mib.json')); print($mib, {k:h[k] for k in ('save_spz_first_s','save_spz_s','load_spz_s')})"
done
