#!/usr/bin/env python3
"""rocprofv3 output of tools/all_kernels.py (kernel trace + FETCH_SIZE pass + WRITE_SIZE pass) -> one row per
case: average kernel duration, HBM bytes per call (FETCH_SIZE x 2, WRITE_SIZE x 1: MI355X_MICROARCH.md §HBM and
profiles/r01_pmc_calibration.jsonl) against the algorithmic bytes.

usage: python tools/summarize_all_kernels.py <tag>   (reads gpurun_out/prof_all_{stats,fetch,write}_<tag>, writes profiles/<tag>_all_kernels.json)
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csv.field_size_limit(1 << 30)
OURS = ("spz_encode_kernel", "spz_decode_kernel", "spz_flip_kernel", "spz_cloud_to_ply_rows_kernel", "spz_ply_rows_to_cloud_kernel",
        "spz_decode_gather_kernel", "spz_select_hist_kernel", "spz_select_pick_kernel")


def short(name):
    for k in OURS:
        if k in name:
            return k
    return None


def load(tag, kind, suffix):
    pats = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_all_{kind}_{tag}", "**", f"*_{suffix}.csv"), recursive=True)
    if not pats:
        return None
    with open(max(pats, key=os.path.getmtime)) as f:   # a tag that was run twice: the newest files
        return list(csv.DictReader(f))


def split(dispatches, manifest):
    """dispatches: [(kernel short name, value)] in dispatch order -> per case list of per-call sums."""
    out, pos = [], 0
    for case in manifest["cases"]:
        calls = []
        for _ in range(case["calls"]):
            total = 0.0
            per_kernel = {}
            for name, count in case["kernels"]:
                for _k in range(count):
                    assert pos < len(dispatches) and dispatches[pos][0] == name, (case["case"], name, dispatches[pos] if pos < len(dispatches) else None)
                    total += dispatches[pos][1]
                    per_kernel[name] = per_kernel.get(name, 0.0) + dispatches[pos][1]
                    pos += 1
            calls.append((total, per_kernel))
        out.append(calls)
    assert pos == len(dispatches), (pos, len(dispatches))
    return out


def main():
    tag = sys.argv[1]
    manifest = json.load(open(os.path.join(ROOT, "gpurun_out", f"all_kernels_manifest_{tag}.json")))
    res = {"tag": tag, "launches_per_case": manifest["launches"], "hbm_peak_GBps": 8000, "rows": []}
    trace = load(tag, "stats", "kernel_trace")
    dur = None
    if trace:
        rows = sorted((r for r in trace if short(r["Kernel_Name"])), key=lambda r: int(r["Start_Timestamp"]))
        dur = split([(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows], manifest)
    traffic = {}
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        cc = load(tag, kind, "counter_collection")
        if cc:
            rows = sorted((r for r in cc if short(r["Kernel_Name"]) and r["Counter_Name"] == counter), key=lambda r: int(r["Dispatch_Id"]))
            traffic[counter] = split([(short(r["Kernel_Name"]), float(r["Counter_Value"]) * 1024.0) for r in rows], manifest)
    sq = {}
    cc = load(tag, "sq", "counter_collection")
    if cc:
        for counter in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"):
            rows = sorted((r for r in cc if short(r["Kernel_Name"]) and r["Counter_Name"] == counter), key=lambda r: int(r["Dispatch_Id"]))
            sq[counter] = split([(short(r["Kernel_Name"]), float(r["Counter_Value"])) for r in rows], manifest)

    def sq_fields(i, name=None):
        if len(sq) < 4:
            return {}
        tot = {c: sum((call[1][name] if name else call[0]) for call in sq[c][i]) for c in sq}
        if tot["SQ_WAVES"] <= 0:
            return {}
        return {"valu_insts_per_wave": round(tot["SQ_INSTS_VALU"] / tot["SQ_WAVES"], 1),
                "wait_fraction_of_wave_cycles": round(tot["SQ_WAIT_ANY"] / max(1.0, tot["SQ_WAVE_CYCLES"]), 3)}

    for i, case in enumerate(manifest["cases"]):
        if case.get("skip"):
            continue
        if case.get("per_kernel_algorithmic"):   # one row per kernel of the case (encode / decode of a pair)
            for name, alg in case["per_kernel_algorithmic"].items():
                row = {"case": f"{name.replace('spz_', '').replace('_kernel', '')} {case['case']}", "kernels": [name],
                       "algorithmic_bytes_per_call": alg, "timed_as": "encode/decode alternating (steady state of the pair)"}
                if dur:
                    us = [c[1][name] for c in dur[i]]
                    row["kernel_us_avg"] = round(sum(us) / len(us), 2)
                    row["kernel_us_min"] = round(min(us), 2)
                    row["algorithmic_GBps"] = round(alg / (row["kernel_us_avg"] * 1e-6) / 1e9, 1)
                    row["frac_of_8TBps"] = round(row["algorithmic_GBps"] / 8000, 3)
                if "FETCH_SIZE" in traffic and "WRITE_SIZE" in traffic:
                    rd = 2.0 * sum(c[1][name] for c in traffic["FETCH_SIZE"][i]) / len(traffic["FETCH_SIZE"][i])
                    wr = sum(c[1][name] for c in traffic["WRITE_SIZE"][i]) / len(traffic["WRITE_SIZE"][i])
                    row["hbm_read_bytes_per_call"] = round(rd)
                    row["hbm_write_bytes_per_call"] = round(wr)
                    row["traffic_over_algorithmic"] = round((rd + wr) / alg, 3)
                row.update(sq_fields(i, name))
                res["rows"].append(row)
            continue
        row = {"case": case["case"], "kernels": sorted({k for k, _ in case["kernels"]}), "algorithmic_bytes_per_call": case["algorithmic_bytes"]}
        if case.get("note"):
            row["note"] = case["note"]
        if dur:
            us = [c[0] for c in dur[i]]
            row["kernel_us_avg"] = round(sum(us) / len(us), 2)
            row["kernel_us_min"] = round(min(us), 2)
            row["algorithmic_GBps"] = round(case["algorithmic_bytes"] / (row["kernel_us_avg"] * 1e-6) / 1e9, 1)
            row["frac_of_8TBps"] = round(row["algorithmic_GBps"] / 8000, 3)
            if len(row["kernels"]) > 1:
                per = {}
                for _, pk in dur[i]:
                    for k, v in pk.items():
                        per[k] = per.get(k, 0.0) + v / len(dur[i])
                row["kernel_us_avg_by_kernel"] = {k: round(v, 2) for k, v in per.items()}
        if "FETCH_SIZE" in traffic and "WRITE_SIZE" in traffic:
            rd = 2.0 * sum(c[0] for c in traffic["FETCH_SIZE"][i]) / len(traffic["FETCH_SIZE"][i])
            wr = sum(c[0] for c in traffic["WRITE_SIZE"][i]) / len(traffic["WRITE_SIZE"][i])
            row["hbm_read_bytes_per_call"] = round(rd)
            row["hbm_write_bytes_per_call"] = round(wr)
            row["traffic_over_algorithmic"] = round((rd + wr) / case["algorithmic_bytes"], 3)
        row.update(sq_fields(i))
        res["rows"].append(row)
    res["event_rows_of_the_stats_run"] = manifest.get("event_rows")
    out = os.path.join(ROOT, "profiles", f"{tag}_all_kernels.json")
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    box = [r for r in (manifest.get("event_rows") or []) if r["case"].startswith("box reference")]
    if box:
        res["box_reference"] = box[0]
        print(f"box reference copy: {box[0]['GBps']} GB/s")
    for r in res["rows"]:
        print(f"{r['case'][:58]:58s} {r.get('kernel_us_avg', 0):9.1f} us  {r.get('frac_of_8TBps', 0):5.3f}  traffic/alg {r.get('traffic_over_algorithmic', '-')}  valu/wave {r.get('valu_insts_per_wave', '-')}")


if __name__ == "__main__":
    main()
