#!/usr/bin/env python3
"""gunzip of ordinary single-stream members (what the reference writes): the parallel reader against the
serial one (libdeflate / zlib), per thread count.  Host-only; prints one JSON line per input."""
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import spz_amd.spz as spz  # noqa: E402


def main():
    n_mb = int(sys.argv[1]) if len(sys.argv) > 1 else 130
    rng = np.random.default_rng(3)
    # an SH3 stream look-alike: 9 B positions, 11 B small sections, 45 B bucketed sh per point
    pts = n_mb * 1_000_000 // 65
    pos = rng.integers(0, 256, 9 * pts, dtype=np.uint8)
    mid = rng.integers(0, 256, 11 * pts, dtype=np.uint8)
    sh = np.clip(np.round(rng.normal(128, 20, 45 * pts) / 8) * 8, 0, 255).astype(np.uint8)
    raw = pos.tobytes() + mid.tobytes() + sh.tobytes()
    os.environ["SPZ_AMD_GZIP_EXACT_THREADS"] = "32"
    gz = spz._compress_gzipped(raw)
    res = {"raw_MB": len(raw) / 1e6, "gz_MB": len(gz) / 1e6, "host_cores": os.cpu_count()}
    for threads in (1, 4, 8, 16, 32, 64):
        os.environ["SPZ_AMD_GUNZIP_THREADS"] = str(threads)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            back = spz._decompress_gzipped(gz)
            best = min(best, time.perf_counter() - t0)
        assert back == raw
        res[f"gunzip_{threads}_threads_s"] = round(best, 4)
    t0 = time.perf_counter(); zlib.decompress(gz, 31); res["python_zlib_s"] = round(time.perf_counter() - t0, 4)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
