#!/usr/bin/env python3
"""HBM rate of the .ply row shuffles on device-resident data.
Algorithmic bytes per Gaussian: row (17+3*shDim)*4 + cloud (14+3*shDim)*4; SH3 = 248 + 236 = 484 B."""
import ctypes as C
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from spz_amd import abi
    from spz_amd.synth import FIELDS, make_cloud_torch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    deg, shd = 3, 15
    dev = torch.device("cuda:0")
    L = abi.load_library()
    if os.environ.get('SPZ_PLY_LIB'):
        L = abi.bind(C.CDLL(os.environ['SPZ_PLY_LIB']))
    cloud = make_cloud_torch(n, deg, 3, dev)
    out = {k: torch.empty_like(cloud[k]) for k in FIELDS}
    D = 17 + 3 * shd
    rows = torch.empty(n * D, dtype=torch.float32, device=dev)
    pin = abi.CloudPtrs(*[cloud[k].data_ptr() for k in FIELDS])
    pout = abi.CloudPtrs(*[out[k].data_ptr() for k in FIELDS])
    cols = abi.PlyColumns()
    L.spz_amd_ply_default_columns(shd, C.byref(cols))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tr, tc = [], []
    for i in range(23):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        abi.check(L.spz_amd_cloud_to_ply_rows_device(C.byref(pin), n, shd, 4, rows.data_ptr(), s), "rows")
        e[1].record()
        abi.check(L.spz_amd_ply_rows_to_cloud_device(rows.data_ptr(), n, C.byref(cols), 4, C.byref(pout), s), "cloud")
        e[2].record()
        torch.cuda.synchronize()
        if i >= 3:
            tr.append(e[0].elapsed_time(e[1]))
            tc.append(e[1].elapsed_time(e[2]))
    same = all(torch.equal(out[k].view(torch.int32), cloud[k].view(torch.int32)) for k in FIELDS)
    gb = n * (D + 14 + 3 * shd) * 4 / 1e9
    for name, t in (("spz_cloud_to_ply_rows_kernel", tr), ("spz_ply_rows_to_cloud_kernel", tc)):
        med = statistics.median(t)
        print(json.dumps({"kernel": name, "points": n, "sh_dim": shd, "ms_med": round(med, 4), "ms_min": round(min(t), 4),
                          "algorithmic_GB": round(gb, 3), "GBps_med": round(gb / (med * 1e-3), 1),
                          "frac_of_8TBps": round(gb / (med * 1e-3) / 8000, 3), "round_trip_bit_identical": same}))


if __name__ == "__main__":
    main()
