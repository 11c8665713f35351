#!/usr/bin/env python3
"""Thread scaling of the byte-identical gzip writer on a stream-shaped input (default: the 650 MB of a 10 M-point SH3
cloud), with the writer's own per-phase laps (SPZ_AMD_EXACT_GZIP_TIMING).  Host only.  One JSON line per thread count.
  python tools/gzip_scaling.py [points] [threads ...]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import sys, time, zlib, hashlib
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tools")
import numpy as np
import spz_amd.spz as spz
from gzip_campaign import stream_like
data = stream_like(int(sys.argv[2]), 3, np.random.default_rng(5))
t0 = time.perf_counter(); out = spz._compress_gzipped_exact(data, int(sys.argv[3]), 32, 0); t = time.perf_counter() - t0
t0 = time.perf_counter(); out2 = spz._compress_gzipped_exact(data, int(sys.argv[3]), 32, 0); t2 = time.perf_counter() - t0
print("RESULT", len(data), len(out) if out else -1, round(min(t, t2), 3), hashlib.sha256(out).hexdigest()[:16] if out else "-")
"""


def main():
    points = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    threads = [int(x) for x in sys.argv[2:]] or [8, 16, 32, 64, 128]
    for t in threads:
        if t > (os.cpu_count() or 1):
            continue
        env = dict(os.environ, SPZ_AMD_EXACT_GZIP_TIMING="1", SPZ_AMD_NO_HIP_PRELOAD="1")
        r = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(points), str(t)], capture_output=True, text=True, env=env, timeout=900)
        if r.returncode != 0:
            print(json.dumps({"threads": t, "error": r.stderr[-400:]}))
            continue
        laps = {}
        for line in r.stderr.splitlines():
            if line.startswith("[exactgz]"):
                parts = line.split()
                laps.setdefault(parts[1], []).append(float(parts[2]))
        res = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()
        print(json.dumps({"threads": t, "host_cores": os.cpu_count(), "input_bytes": int(res[1]), "member_bytes": int(res[2]),
                          "best_of_2_s": float(res[3]), "member_sha256_16": res[4],
                          "laps_last_call_s": {k: v[-1] for k, v in laps.items()}}), flush=True)


if __name__ == "__main__":
    main()
