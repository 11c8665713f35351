#!/bin/bash
# the LZ stage's kernels of one saveSpz (10 M SH3, tables and matches NOT beside the upload, so that their durations are
# their own): rocprofv3 kernel statistics, then one counter pass for lz_match_kernel's lane utilisation and load counts
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-lzm}
mkdir -p $O
cd $R
export TMPDIR=/tmp
export SPZ_AMD_GZIP_OVERLAP=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lzm_$TAG -o run -- ./spz_amd/bin/host_bench 10000000 3 2 1 > $O/lzm_$TAG.log 2>&1 || { echo "rocprofv3 failed"; tail -n 8 $O/lzm_$TAG.log; exit 3; }
f=$(find $O/prof_lzm_$TAG -name '*kernel_stats.csv' | head -n 1)
[ -n "$f" ] && cp "$f" $O/lzm_kernel_stats_$TAG.csv && grep -E "Name|lz_|inf_" $O/lzm_kernel_stats_$TAG.csv | cut -c1-160
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/prof_lzm_pmc_$TAG -o run -- ./spz_amd/bin/host_bench 10000000 3 1 1 > $O/lzm_pmc_$TAG.log 2>&1 || { echo "pmc pass failed"; tail -n 8 $O/lzm_pmc_$TAG.log; exit 4; }
c=$(find $O/prof_lzm_pmc_$TAG -name '*counter_collection.csv' | head -n 1)
[ -n "$c" ] && python3 - "$c" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if "lz_" in k or "inf_" in k:
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(k[:60], {a: int(b) for a, b in v.items()})
PY
