#!/usr/bin/env python3
"""Every kernel of libspz_amd.so on device-resident data, a few launches each, in a fixed order — the program
that rocprofv3 wraps for the per-kernel rows of profiles/ (tools/gpu_profile_all.sh):

  --kernel-trace --stats            -> average duration per kernel and case
  --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -> HBM bytes per launch

Writes a manifest (cases in execution order, launches per case, algorithmic bytes per launch) that
tools/summarize_all_kernels.py uses to split the dispatch list, and prints HIP-event timings of its own."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=6)
    ap.add_argument("--cases", default="", help="comma-separated substrings: run only the cases whose name contains one")
    ap.add_argument("--manifest", default=os.path.join(ROOT, "gpurun_out", "all_kernels_manifest.json"))
    a = ap.parse_args()
    import torch

    from spz_amd import abi, device as D
    from spz_amd.synth import FIELDS, make_cloud_torch
    dev = torch.device("cuda:0")
    L = abi.load_library()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    manifest, rows = [], []

    only = [c for c in a.cases.split(",") if c]

    def timed(case, kernels, alg_bytes, fn, note="", per_kernel_alg=None):
        """kernels: list of (name substring, dispatches per call) in dispatch order."""
        if only and not any(c in case for c in only):
            return
        fn()
        torch.cuda.synchronize()
        manifest.append({"case": case + " (warm-up)", "kernels": kernels, "calls": 1, "algorithmic_bytes": alg_bytes, "skip": True})
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        for _ in range(a.launches):
            fn()
        e[1].record()
        torch.cuda.synchronize()
        ms = e[0].elapsed_time(e[1]) / a.launches
        manifest.append({"case": case, "kernels": kernels, "calls": a.launches, "algorithmic_bytes": alg_bytes, "note": note,
                         "per_kernel_algorithmic": per_kernel_alg})
        r = {"case": case, "ms_per_call_events": round(ms, 4), "algorithmic_GB": round(alg_bytes / 1e9, 4),
             "GBps": round(alg_bytes / ms / 1e6, 1), "frac_of_8TBps": round(alg_bytes / ms / 1e6 / 8000, 3)}
        rows.append(r)
        print(json.dumps(r), flush=True)

    # what this box's memory system does on the guide's reference shape (a device-to-device copy of 1 GiB): boxes of
    # the pool differ by more than 10 % on the same binary, so every row is to be read next to this figure
    src = torch.empty(1 << 30, dtype=torch.uint8, device=dev).fill_(7)
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(10):
        dst.copy_(src)
    e[1].record()
    torch.cuda.synchronize()
    box = {"case": "box reference: torch device-to-device copy of 1 GiB (read + write)", "ms_per_call_events": round(e[0].elapsed_time(e[1]) / 10, 4),
           "GBps": round(2 * (1 << 30) / (e[0].elapsed_time(e[1]) / 10) / 1e6, 1)}
    print(json.dumps(box), flush=True)
    rows.append(box)
    del src, dst
    bpp = {0: 76, 1: 121, 2: 196, 3: 301}
    for n, deg, ver, tag in ((10_000_000, 3, 3, "10M sh3 v3"), (10_000_000, 0, 3, "10M sh0 v3"), (10_000_000, 1, 3, "10M sh1 v3"),
                             (1_000_000, 0, 2, "cfg2 1M sh0 v2")):
        cloud = make_cloud_torch(n, deg, 3, dev)
        out = {k: torch.empty_like(cloud[k]) for k in FIELDS}
        lay = abi.stream_layout(n, deg, ver)
        stream = torch.empty(lay.total_bytes, dtype=torch.uint8, device=dev)
        pin = abi.CloudPtrs(*[cloud[k].data_ptr() for k in FIELDS])
        pout = abi.CloudPtrs(*[out[k].data_ptr() for k in FIELDS])
        hdr = abi.Header(ver, n, deg, 12, 0, 0)
        alg = n * (bpp[deg] - (1 if ver == 2 else 0))
        def pair():
            abi.check(L.spz_amd_encode_device(C.byref(pin), n, deg, 0, 6, ver, stream.data_ptr(), stream.numel(), s), "enc")
            abi.check(L.spz_amd_decode_device(stream.data_ptr(), stream.numel(), C.byref(hdr), 6, C.byref(pout), s), "dec")
        # encode and decode alternate, as in bench.py's step: what one kernel leaves in the write-back caches is
        # paid by the next, so each is timed in the pair's own steady state (profiles/r01_tune_h_*)
        timed(f"{tag}", [("spz_encode_kernel", 1), ("spz_decode_kernel", 1)], 2 * alg, pair,
              per_kernel_alg={"spz_encode_kernel": alg, "spz_decode_kernel": alg})
        if deg == 3:
            d = 45
            timed("convertCoordinates 10M sh3 (standalone flip pass)", [("spz_flip_kernel", 1)], n * (3 + 4 + d) * 8,
                  lambda: abi.check(L.spz_amd_convert_coordinates_device(out["positions"].data_ptr(), out["rotations"].data_ptr(),
                                                                         out["sh"].data_ptr(), n, deg, 4, 6, s), "flip"))
            # .ply rows <-> cloud
            shd, Dr = 15, 17 + 45
            prow = torch.empty(n * Dr, dtype=torch.float32, device=dev)
            cols = abi.PlyColumns()
            L.spz_amd_ply_default_columns(shd, C.byref(cols))
            alg_ply = n * (Dr + 14 + 45) * 4
            timed("cloud -> .ply rows 10M sh3", [("spz_cloud_to_ply_rows_kernel", 1)], alg_ply,
                  lambda: abi.check(L.spz_amd_cloud_to_ply_rows_device(C.byref(pin), n, shd, 4, prow.data_ptr(), s), "rows"))
            timed(".ply rows -> cloud 10M sh3", [("spz_ply_rows_to_cloud_kernel", 1)], alg_ply,
                  lambda: abi.check(L.spz_amd_ply_rows_to_cloud_device(prow.data_ptr(), n, C.byref(cols), 4, C.byref(pout), s), "cloud"))
            del prow
            # random access: 10 M random indices out of the packed stream
            idx = torch.randint(0, n, (n,), device=dev, dtype=torch.int32)
            timed("gather decode, 10M random indices of a 10M sh3 stream", [("spz_decode_gather_kernel", 1)], n * 301 + n * 4,
                  lambda: D.decode_gather(stream, hdr, idx, 0, out=out),
                  note="algorithmic = 65 B read + 236 B written + 4 B index per point.  The attribute-major format puts a point's "
                       "65 bytes into six sections, so a random point touches ~7 sectors of 64 B (448 B): inherent to the format. "
                       "The x2 correction of FETCH_SIZE is calibrated for streaming reads only; for these scattered sector reads "
                       "the true read traffic lies between FETCH_SIZE x1 and x2")
            del idx
            # medianVolume's selection: 4 histogram passes + 4 one-wave picks
            ws = torch.empty(abi.MEDIAN_WORKSPACE_BYTES, dtype=torch.uint8, device=dev)
            med = torch.empty(1, dtype=torch.float32, device=dev)
            timed("medianVolume selection 10M (4 histogram passes)", [("spz_select_hist_kernel", 1), ("spz_select_pick_kernel", 1)] * 4,
                  n * 48, lambda: abi.check(L.spz_amd_median_scale_sum_device(cloud["scales"].data_ptr(), n, ws.data_ptr(),
                                                                              med.data_ptr(), s), "median"),
                  note="algorithmic = 4 passes x 12 B per point")
            if only and not any(c in "medianVolume" for c in only):
                continue
            want = (cloud["scales"].view(-1, 3)[:, 0] + cloud["scales"].view(-1, 3)[:, 1] + cloud["scales"].view(-1, 3)[:, 2]).sort().values[n // 2]
            assert float(med[0]) == float(want), (float(med[0]), float(want))
            # the same on a distribution that puts every sum into ONE first-pass bin (sums in [-6, -4))
            narrow = torch.empty(3 * n, dtype=torch.float32, device=dev).uniform_(-2.0, -1.34)
            timed("medianVolume selection 10M, all sums in one exponent bin", [("spz_select_hist_kernel", 1), ("spz_select_pick_kernel", 1)] * 4,
                  n * 48, lambda: abi.check(L.spz_amd_median_scale_sum_device(narrow.data_ptr(), n, ws.data_ptr(), med.data_ptr(), s), "median"))
            v = narrow.view(-1, 3)
            want = ((v[:, 0] + v[:, 1]) + v[:, 2]).sort().values[n // 2]
            assert float(med[0]) == float(want), (float(med[0]), float(want))
            del narrow, v
        del cloud, out, stream
    os.makedirs(os.path.dirname(a.manifest), exist_ok=True)
    with open(a.manifest, "w") as f:
        json.dump({"launches": a.launches, "cases": manifest, "event_rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
