#!/usr/bin/env python3
"""min / median / max of the decode times per (kind, experiment) of a placement_scan.hip run."""
import json
import statistics
import sys
from collections import defaultdict

rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
for r in rows:
    if r["exp"] == "row":
        print(f"{r['stream_chunk']:2d} {r['ptr']} " + " ".join(f"{int(round(v * 1000)) - 400:3d}" for v in r["dec_il_ms"]))
for r in rows:
    if r["exp"] == "bw":
        print(r["pass"], r["chunk"], r["ptr"], r["write_GBps"], r["read_GBps"])
for r in rows:
    if r["exp"] == "pick":
        print("pick", r["trial"], "ref", r["ref_il_ms"], "cands", " ".join(f"{x:.3f}" for x in r["cand_il_ms"]))
rows = [r for r in rows if r["exp"] != "pick"]
vr = [r for r in rows if r["exp"] == "variant"]
if vr:
    places = sorted({r["place"] for r in vr}, key=lambda x: (x != "same", len(x), x))
    print("variant      " + " ".join(f"{p:>13s}" for p in places) + "   (dec ms / enc ms)")
    for name in dict.fromkeys(r["variant"] for r in vr):
        cells = {r["place"]: r for r in vr if r["variant"] == name}
        print(f"{name:12s} " + " ".join(f"{cells[p]['dec_ms']:.3f}/{cells[p]['enc_ms']:.3f}  " if p in cells else " " * 13 for p in places))
rows = [r for r in rows if r["exp"] not in ("row", "bw", "variant")]
for r in rows:
    if r["exp"] == "fresh":
        print(r["kind"], r["param_mib"], r["dec_seq_ms"], r["dec_il_ms"], r["enc_ms"])
g = defaultdict(list)
for r in rows:
    g[(r["kind"], r["exp"])].append(r)
print(f"{'kind':8s} {'exp':12s} {'n':>3s}  seq min/med/max        il min/med/max         enc med")
for (kind, exp), v in g.items():
    s = sorted(x["dec_seq_ms"] for x in v)
    i = sorted(x["dec_il_ms"] for x in v)
    e = statistics.median(x["enc_ms"] for x in v)
    print(f"{kind:8s} {exp:12s} {len(v):3d}  {s[0]:.3f} {statistics.median(s):.3f} {s[-1]:.3f}    {i[0]:.3f} {statistics.median(i):.3f} {i[-1]:.3f}    {e:.3f}")
