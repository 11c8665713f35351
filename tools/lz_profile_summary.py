#!/usr/bin/env python3
"""Per-kernel durations (rocprofv3 kernel_stats csv) and counter sums (counter_collection csv) of the container stage's kernels.
  python tools/lz_profile_summary.py gpurun_out/lzm_kernel_stats_X.csv [gpurun_out/prof_lzm_pmc_X/run_counter_collection.csv]"""
import collections
import csv
import sys


def short(name):
    toks = [t for t in name.replace("(", " ").replace("<", " <").split() if "lz_" in t or "inf_" in t]
    if not toks:
        return None
    base = toks[0].split("::")[-1]
    return base + ("<1>" if "<1>" in name.split("(")[0] else "<0>" if "<0>" in name.split("(")[0] else "")


for r in csv.DictReader(open(sys.argv[1])):
    n = short(r["Name"])
    if n:
        print(f"{n:28s} calls {r['Calls']:>4} total {float(r['TotalDurationNs'])/1e6:8.2f} ms avg {float(r['AverageNs'])/1e6:8.3f}")
if len(sys.argv) > 2:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(sys.argv[2])):
        n = short(r["Kernel_Name"])
        if n:
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(k, {a: int(b) for a, b in sorted(v.items())})
