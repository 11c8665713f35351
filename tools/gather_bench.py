#!/usr/bin/env python3
"""Random-access decode (spz_amd_decode_gather_device): time to pull K random points out of a packed
10 M-point SH3 stream that stays in HBM, against the bytes it must move (65 B read, 236 B written per point)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from spz_amd import device as D  # noqa: E402
from spz_amd.synth import make_cloud_torch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n, deg = 10_000_000, 3
    cloud = make_cloud_torch(n, deg, 3, dev)
    stream = D.encode(cloud, n, deg)
    hdr = D.make_header(n, deg)
    del cloud
    for k in (1_000, 100_000, 1_000_000, 10_000_000):
        for order in ("random", "sorted"):
            idx = torch.randint(0, n, (k,), device=dev, dtype=torch.int32)
            if order == "sorted":
                idx = idx.sort().values
            out = D.alloc_cloud(k, deg, dev)
            D.decode_gather(stream, hdr, idx, 0, out=out)
            torch.cuda.synchronize()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            reps = 10
            e[0].record()
            for _ in range(reps):
                D.decode_gather(stream, hdr, idx, 0, out=out)
            e[1].record()
            torch.cuda.synchronize()
            ms = e[0].elapsed_time(e[1]) / reps
            print(json.dumps({"points": k, "order": order, "ms": round(ms, 4), "M_points_per_s": round(k / ms / 1e3, 1),
                              "algorithmic_GBps": round(k * 301 / ms / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
