#!/usr/bin/env python3
"""compressGzipped of a 10 M-point SH3 stream (650 MB) with the parse on the device vs on the host's threads:
seconds, identity of the two members, per-stage laps on stderr (SPZ_AMD_LZ_TIMING, SPZ_AMD_EXACT_GZIP_TIMING)."""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import spz_amd.spz as spz  # noqa: E402
from spz_amd.synth import make_cloud_numpy  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    c = make_cloud_numpy(n, 3, 3)
    g = spz.GaussianCloud()
    g.sh_degree = 3
    for k in ("positions", "scales", "rotations", "alphas", "colors", "sh"):
        setattr(g, k, c[k])
    o = spz.PackOptions()
    o.from_coord = spz.RDF
    raw = spz._pack_to_stream(g, o)
    del c, g
    out = {"points": n, "stream_bytes": len(raw), "usable_cpus": spz._effective_cpu_count()}
    os.environ["SPZ_AMD_LZ_TIMING"] = "1"
    os.environ["SPZ_AMD_EXACT_GZIP_TIMING"] = "1"
    for mode in ("1", "parse_only", "0"):
        os.environ["SPZ_AMD_GZIP_DEVICE"] = "0" if mode == "0" else "1"
        os.environ["SPZ_AMD_GZIP_DEVICE_HUFFMAN"] = "0" if mode == "parse_only" else "1"
        best = None
        for r in range(reps if mode != "0" else 1):
            before = spz._device_gzip_parse_count()
            t = time.perf_counter()
            m = spz._compress_gzipped(raw)
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
            used = spz._device_gzip_parse_count() - before
            print(f"[bench] device={mode} run {r}: {dt:.3f} s, device parses {used}", file=sys.stderr, flush=True)
        key = {"1": "device_parse_and_huffman", "parse_only": "device_parse", "0": "host_threads"}[mode]
        out[key + "_s"] = round(best, 4)
        out[key + "_member_sha256_16"] = hashlib.sha256(m).hexdigest()[:16]
        out[key + "_member_bytes"] = len(m)
        if mode == "1":
            out["device_parse_used"] = bool(used)
    out["members_identical"] = (out["device_parse_member_sha256_16"] == out["host_threads_member_sha256_16"] ==
                                out["device_parse_and_huffman_member_sha256_16"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
