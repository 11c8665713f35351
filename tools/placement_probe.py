#!/usr/bin/env python3
"""Decode time against WHERE the buffers lie.  Many random placements in one process (fresh allocations with random
spacers in between, so that the device addresses vary), sequential and interleaved grid; every line carries the device
addresses, for an offline look at which address bits decide.  Finding so far (profiles/r02_placement_*.jsonl): the
same binary on the same box runs its sh3 decode at 0.46 ms or at 0.58 ms depending on nothing but the addresses the
allocator happened to return."""
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
    from spz_amd.synth import FIELDS, make_cloud_numpy, floats_per_point
    dev = torch.device("cuda:0")
    n, deg = 10_000_000, 3
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    names = (sys.argv[2] if len(sys.argv) > 2 else "seq,policy").split(",")
    fresh_cloud = len(sys.argv) > 3 and sys.argv[3] == "fresh"   # re-allocate the input arrays too (placement then varies within a process)
    libs = {name: abi.bind(C.CDLL(os.path.join(ROOT, "build", "variants", f"libspz_amd_{name}.so"))) for name in names}
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    lay = abi.stream_layout(n, deg, 3)
    hdr = abi.Header(3, n, deg, 12, 0, 0)
    host = make_cloud_numpy(n, deg, 3)
    sizes = {k: n * floats_per_point(k, deg) * 4 for k in FIELDS}
    cloud = {k: torch.from_numpy(host[k]).to(dev) for k in FIELDS}
    pin = abi.CloudPtrs(*[cloud[k].data_ptr() for k in FIELDS])
    g = torch.Generator().manual_seed(7)
    spacers = []
    for trial in range(trials):
        if fresh_cloud:
            del cloud
            torch.cuda.empty_cache()
            cloud = {k: torch.from_numpy(host[k]).to(dev) for k in FIELDS}
            pin = abi.CloudPtrs(*[cloud[k].data_ptr() for k in FIELDS])

        def spacer():
            spacers.append(torch.empty(int(torch.randint(1, 200, (1,), generator=g)) << 20, dtype=torch.uint8, device=dev))
        out = {}
        for k in FIELDS:
            spacer()
            out[k] = torch.empty(sizes[k] // 4, dtype=torch.float32, device=dev)
        spacer()
        stream = torch.empty(lay.total_bytes, dtype=torch.uint8, device=dev)
        pout = abi.CloudPtrs(*[out[k].data_ptr() for k in FIELDS])
        r = {"trial": trial, "stream_ptr": stream.data_ptr(), "out_ptr": {k: out[k].data_ptr() for k in FIELDS},
             "cloud_ptr": {k: cloud[k].data_ptr() for k in FIELDS}}
        for name, L in libs.items():
            enc = lambda: abi.check(L.spz_amd_encode_device(C.byref(pin), n, deg, 0, 6, 3, stream.data_ptr(), stream.numel(), s), "enc")
            dec = lambda: abi.check(L.spz_amd_decode_device(stream.data_ptr(), stream.numel(), C.byref(hdr), 6, C.byref(pout), s), "dec")
            enc(); dec(); enc(); dec()
            te, td = [], []
            for _ in range(7):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record(); enc(); e[1].record(); dec(); e[2].record()
                torch.cuda.synchronize()
                te.append(e[0].elapsed_time(e[1])); td.append(e[1].elapsed_time(e[2]))
            r[f"{name}_enc_ms"] = round(statistics.median(te), 4)
            r[f"{name}_dec_ms"] = round(statistics.median(td), 4)
        print(json.dumps(r), flush=True)
        del out, stream
        if len(spacers) > 60:
            del spacers[:30]
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
