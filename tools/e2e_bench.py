#!/usr/bin/env python3
"""End-to-end saveSpz / loadSpz through the Python module (host arrays in, .spz bytes out): where the
wall-clock goes once the quantise step is on the GPU, and what the opt-in parallel gzip buys."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import spz_amd.spz as spz  # noqa: E402
from spz_amd.synth import FIELDS, make_cloud_numpy  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    deg = 3
    c = make_cloud_numpy(n, deg, 3)
    g = spz.GaussianCloud()
    g.sh_degree = deg
    for k in FIELDS:
        setattr(g, k, c[k])
    o = spz.PackOptions()
    o.from_coord = spz.RDF
    res = {"points": n, "sh_degree": deg, "host_cores": os.cpu_count()}
    t0 = time.perf_counter(); raw = spz._pack_to_stream(g, o); res["pack_to_stream_s"] = round(time.perf_counter() - t0, 4)
    t0 = time.perf_counter(); raw = spz._pack_to_stream(g, o); res["pack_to_stream_warm_s"] = round(time.perf_counter() - t0, 4)
    res["stream_bytes"] = len(raw)
    # the same two calls timed at the C++ boundary (fresh std::vectors, no copy into a Python bytes object: that copy,
    # 650 MB into freshly mapped memory for a 10 M-point cloud, is what the two figures above carry on top)
    u0 = spz.UnpackOptions()
    u0.to_coord = spz.RDF
    best = [min(x) for x in zip(*[spz._pack_unpack_seconds(g, o, u0)[:2] for _ in range(3)])]
    res["cpp_pack_to_stream_s"], res["cpp_unpack_from_stream_s"] = round(best[0], 4), round(best[1], 4)
    # the reference's container (byte-identical .spz): zlib itself, then the exact multi-threaded writer
    os.environ["SPZ_AMD_GZIP_EXACT_THREADS"] = "1"
    t0 = time.perf_counter(); z_ref = spz._compress_gzipped(raw); res["gzip_zlib_single_stream_s"] = round(time.perf_counter() - t0, 3)
    del os.environ["SPZ_AMD_GZIP_EXACT_THREADS"]
    t0 = time.perf_counter(); z_exact = spz._compress_gzipped(raw); res["gzip_exact_parallel_default_s"] = round(time.perf_counter() - t0, 3)
    res["gzip_exact_parallel_identical_to_zlib"] = bool(z_exact == z_ref)
    res["gzip_reference_container_bytes"] = len(z_ref)
    t0 = time.perf_counter(); b0 = spz._save_spz_bytes(g, o); res["save_spz_total_default_s"] = round(time.perf_counter() - t0, 3)
    res["save_spz_default_identical_to_zlib_container"] = bool(b0 == z_ref)
    # the opt-in container (independent pieces + index: different bytes, parallel loads)
    for threads in (8, 32, 128):
        if threads > (os.cpu_count() or 1):
            continue
        t0 = time.perf_counter()
        z = spz._compress_gzipped_parallel(raw, threads)
        res[f"gzip_indexed_{threads}_threads_s"] = round(time.perf_counter() - t0, 3)
        res[f"gzip_indexed_{threads}_threads_bytes"] = len(z)
    # three readers: the piece-parallel one (z carries the index), libdeflate and zlib on the reference's
    # single-stream member (z1); the zlib-only figure is taken with Python's zlib, the same library
    import zlib
    z1 = z_ref
    t0 = time.perf_counter(); back = spz._decompress_gzipped(z); res["gunzip_indexed_parallel_s"] = round(time.perf_counter() - t0, 3)
    assert back == raw
    t0 = time.perf_counter(); back = spz._decompress_gzipped(z1); res["gunzip_single_stream_s"] = round(time.perf_counter() - t0, 3)
    assert back == raw
    t0 = time.perf_counter(); back = zlib.decompress(z1, 31); res["gunzip_single_stream_zlib_s"] = round(time.perf_counter() - t0, 3)
    del back
    u1 = spz.UnpackOptions()
    u1.to_coord = spz.RDF
    t0 = time.perf_counter(); d = spz._load_spz_bytes(z1, u1); res["load_spz_total_single_stream_s"] = round(time.perf_counter() - t0, 3)
    u = spz.UnpackOptions()
    u.to_coord = spz.RDF
    t0 = time.perf_counter(); d = spz._load_spz_bytes(z, u); res["load_spz_total_indexed_s"] = round(time.perf_counter() - t0, 3)
    assert d.num_points == n
    os.environ["SPZ_AMD_GZIP_THREADS"] = "64"
    t0 = time.perf_counter(); b = spz._save_spz_bytes(g, o); res["save_spz_total_indexed_64_threads_s"] = round(time.perf_counter() - t0, 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
