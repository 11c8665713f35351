#!/bin/bash
# round 3: when do the LZ kernels of a saveSpz run relative to its uploads?  rocprofv3 kernel + memory-copy trace of host_bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof_timeline -- $R/spz_amd/bin/host_bench 10000000 3 2 1 > $O/timeline.json 2> $O/timeline.err || { echo "rocprof failed"; tail -n 5 $O/timeline.err; exit 3; }
python3 - $O/prof_timeline <<'PY'
import csv, glob, os, sys
d = sys.argv[1]
kt = max(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
mc = max(glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True), key=os.path.getmtime)
ev = []
for r in csv.DictReader(open(kt)):
    import re
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:40]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + name))
for r in csv.DictReader(open(mc)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "?"))[:24]))
ev.sort()
# the last saveSpz: from the last run of lz_copy... find the last 'spz_encode_kernel' cluster preceding lz kernels
t0 = None
for s, e, n in ev:
    if "lz_copy_kernel" in n:
        t0 = s
# go back to the first H2D copy within 80 ms before the LAST session's first lz_copy
firsts = [s for s, e, n in ev if "lz_copy_kernel" in n]
# sessions: cluster lz_copy starts by gaps > 200 ms
sess = []
for s in firsts:
    if not sess or s - sess[-1][-1] > 200e6:
        sess.append([s])
    else:
        sess[-1].append(s)
start = sess[-1][0] - 30e6
end = start + 200e6
print("timeline of the last saveSpz (ms from 30 ms before its first feed); copies > 2 ms and kernels > 0.5 ms:")
for s, e, n in ev:
    if s < start or s > end:
        continue
    dur = (e - s) / 1e6
    if (n.startswith("C") and dur > 0.5) or (n.startswith("K") and dur > 0.5):
        print(f"  {(s - start) / 1e6:8.2f} .. {(e - start) / 1e6:8.2f}  {dur:7.2f} ms  {n}")
PY
