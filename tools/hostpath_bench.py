#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (spz_amd_encode_host / spz_amd_decode_host):
pageable host arrays in, host stream out, blocking — what the C++ saveSpz/loadSpz layer pays before
gzip.  Reported in DESIGN.md §9; never the bench `value`."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from spz_amd import abi  # noqa: E402
from spz_amd.synth import FIELDS, make_cloud_numpy  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    deg = 3
    L = abi.load_library()
    t0 = time.perf_counter()
    c = make_cloud_numpy(n, deg, 3)
    gen_s = time.perf_counter() - t0
    lay = abi.stream_layout(n, deg, 3)
    stream = np.zeros(lay.total_bytes, np.uint8)
    out = {k: np.zeros_like(c[k]) for k in FIELDS}
    pin = abi.CloudPtrs(*[c[k].ctypes.data for k in FIELDS])
    pout = abi.CloudPtrs(*[out[k].ctypes.data for k in FIELDS])
    res = {"points": n, "sh_degree": deg, "float_bytes": int(sum(c[k].nbytes for k in FIELDS)),
           "stream_bytes": int(lay.total_bytes), "synth_s": round(gen_s, 2)}
    for name, fn in (("encode_host", lambda: L.spz_amd_encode_host(C.byref(pin), n, deg, 0, 6, 3, stream.ctypes.data,
                                                                   stream.size, 0)),
                     ("decode_host", lambda: L.spz_amd_decode_host(stream.ctypes.data, stream.size, 6, C.byref(pout), 0))):
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            rc = fn()
            ts.append(time.perf_counter() - t0)
            assert rc == 0, (name, rc)
        best = min(ts[1:])
        res[name] = {"first_s": round(ts[0], 4), "best_s": round(best, 4), "gaussians_per_s": n / best,
                     "GBps_over_pcie": (res["float_bytes"] + res["stream_bytes"]) / best / 1e9}
    # latency of a tiny call (2 Gaussians): what a session of many small saves/loads pays per call
    c2 = make_cloud_numpy(2, deg, 4)
    lay2 = abi.stream_layout(2, deg, 3)
    s2 = np.zeros(lay2.total_bytes, np.uint8)
    p2 = abi.CloudPtrs(*[c2[k].ctypes.data for k in FIELDS])
    for _ in range(20):
        L.spz_amd_encode_host(C.byref(p2), 2, deg, 0, 6, 3, s2.ctypes.data, s2.size, 0)
    t0 = time.perf_counter()
    for _ in range(500):
        L.spz_amd_encode_host(C.byref(p2), 2, deg, 0, 6, 3, s2.ctypes.data, s2.size, 0)
    res["encode_host_2_points_us"] = round((time.perf_counter() - t0) / 500 * 1e6, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
