#!/bin/bash
# SQ counters of the device inflate's kernels (one pass per counter group)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_FLAT SQ_IFETCH SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/prof_inf_$i -o run --output-format csv -- ./spz_amd/bin/host_bench 10000000 3 1 1 > gpurun_out/inf_pmc_$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections
csv.field_size_limit(1<<30)
for d in sorted(glob.glob("gpurun_out/prof_inf_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "inf_" in k:
                acc[(k.split("(")[0][-40:], r["Counter_Name"])] += float(r["Counter_Value"])
        for (k, c), v in sorted(acc.items()):
            print(f"{k:42s} {c:24s} {v:.4g}")
PY
