#!/bin/bash
# which hardware counters move with the placement kind?  rocprofv3 --pmc passes over tools/placement_pmc.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -c Counter_Name $O/counters_list.txt
grep -i -E "Counter_Name.*(UTCL|TLB|TRANSL)" $O/counters_list.txt | awk '{print $NF}' | sort -u | tr '\n' ' '
echo
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "TCC_BUSY_sum TCC_TAG_STALL_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/prof_place_$i -- python3 $R/tools/placement_pmc.py 8 > $O/placement_pmc_$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -n 3 $O/placement_pmc_$i.log; continue; }
  python3 - "$O/prof_place_$i" "$set" <<'PY'
import csv, glob, sys, os
csv.field_size_limit(1 << 30)
d, names = sys.argv[1], sys.argv[2].split()
cc = max(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
rows = {}
for r in csv.DictReader(open(cc)):
    if "spz_decode_kernel" not in r["Kernel_Name"]:
        continue
    e = rows.setdefault(int(r["Dispatch_Id"]), {"dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
print("decode dispatches (4 per placement):", " | ".join(names))
for k in range(0, len(ids), 4):
    grp = [rows[i] for i in ids[k:k + 4]][1:]
    if not grp:
        continue
    dur = sum(g["dur"] for g in grp) / len(grp)
    vals = [sum(g.get(nm, 0.0) for g in grp) / len(grp) for nm in names]
    print(f"  placement {k // 4}: {dur:7.1f} us  " + "  ".join(f"{v:14.0f}" for v in vals))
PY
done
