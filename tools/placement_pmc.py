#!/usr/bin/env python3
"""Fresh allocations of every array per trial (the case in which one process sees both placement kinds); encode+decode
pairs of the shipped library, a marker print per trial.  Meant to run under rocprofv3 --pmc <counters> --kernel-trace:
tools/gpu_placement_pmc.sh correlates each decode dispatch's duration with the counters."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from spz_amd import abi
    from spz_amd.synth import FIELDS, make_cloud_numpy, floats_per_point
    dev = torch.device("cuda:0")
    n, deg = 10_000_000, 3
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    L = abi.load_library()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    lay = abi.stream_layout(n, deg, 3)
    hdr = abi.Header(3, n, deg, 12, 0, 0)
    host = make_cloud_numpy(n, deg, 3)
    sizes = {k: n * floats_per_point(k, deg) for k in FIELDS}
    for trial in range(trials):
        cloud = {k: torch.empty(sizes[k], dtype=torch.float32, device=dev) for k in FIELDS}
        out = {k: torch.empty(sizes[k], dtype=torch.float32, device=dev) for k in FIELDS}
        stream = torch.empty(lay.total_bytes, dtype=torch.uint8, device=dev)
        for k in FIELDS:
            cloud[k].copy_(torch.from_numpy(host[k]))
        pin = abi.CloudPtrs(*[cloud[k].data_ptr() for k in FIELDS])
        pout = abi.CloudPtrs(*[out[k].data_ptr() for k in FIELDS])
        for _ in range(4):
            abi.check(L.spz_amd_encode_device(C.byref(pin), n, deg, 0, 6, 3, stream.data_ptr(), stream.numel(), s), "enc")
            abi.check(L.spz_amd_decode_device(stream.data_ptr(), stream.numel(), C.byref(hdr), 6, C.byref(pout), s), "dec")
        torch.cuda.synchronize()
        print(json.dumps({"trial": trial, "pairs": 4}), flush=True)
        del cloud, out, stream
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
