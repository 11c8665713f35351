#!/usr/bin/env python3
"""gunzip of an ordinary single-stream member (what the reference writes) holding a real SH3 stream: the
parallel reader against the serial one (libdeflate / zlib), per thread count.  Prints one JSON line."""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    from spz_amd.synth import FIELDS, make_cloud_numpy
    c = make_cloud_numpy(n, 3, 3)
    g = spz.GaussianCloud()
    g.sh_degree = 3
    for k in FIELDS:
        setattr(g, k, c[k])
    raw = spz._pack_to_stream(g, spz.PackOptions())      # the quantise step runs on the GPU
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
