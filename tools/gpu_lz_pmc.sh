#!/bin/bash
# SQ counters of the device gzip writer's kernels (one pass per counter group)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/prof_lz_$i -o run --output-format csv -- ./spz_amd/bin/host_bench 10000000 3 1 1 > gpurun_out/lz_pmc_$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections
csv.field_size_limit(1<<30)
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
for d in sorted(glob.glob("gpurun_out/prof_lz_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            for name in ("lz_match_kernel", "lz_table_kernel", "lz_parse_kernel", "lz_encode_kernel"):
                if name in k:
                    acc[(name, r["Counter_Name"])] += float(r["Counter_Value"])
for (k, c), v in sorted(acc.items()):
    print(f"{k:20s} {c:28s} {v:.4g}")
PY
