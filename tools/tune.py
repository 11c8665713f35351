#!/usr/bin/env python3
"""Build-variant tuner: times encode / decode of the bench workload for several builds of
spz_kernels.hip (launch geometry, non-temporal accesses) in ONE process, interleaved rounds,
HIP-event timing, median + min reported (cdna_hip_programming.md §5.4 rule 24).

  python tools/tune.py build            # in the container: hipcc every variant -> build/variants/
  python tools/tune.py run [--points N] # on the GPU box: time them
"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "build", "variants")

def E(block=256, unroll=4, wc=0, ntl=0, nts=0):
    return {"SPZ_ENC_BLOCK": block, "SPZ_ENC_UNROLL": unroll, "SPZ_ENC_WC": wc, "SPZ_ENC_NTL": ntl, "SPZ_ENC_NTS": nts}


def D(block=256, unroll=4, wc=0, ntl=0, nts=0):
    return {"SPZ_DEC_BLOCK": block, "SPZ_DEC_UNROLL": unroll, "SPZ_DEC_WC": wc, "SPZ_DEC_NTL": ntl, "SPZ_DEC_NTS": nts}


# Round-2 experiments (the round-1 geometry / non-temporal / reverse-order tables are in git history and
# profiles/r01_tune_*.jsonl).  Every variant is the shipped configuration plus the listed macros.
_IL = {"SPZ_ENC_INTERLEAVE": 1, "SPZ_DEC_INTERLEAVE": 1}
VARIANTS = {
    "quat_ieee": {"enc": "plain IEEE divisions, sequential grid", "dec": "same",
                  "defs": {"SPZ_QUAT_FAST": 0, "SPZ_ENC_INTERLEAVE": 0, "SPZ_DEC_INTERLEAVE": 0}},
    "seq": {"enc": "sequential grid (sections one after another)", "dec": "same",
            "defs": {"SPZ_ENC_INTERLEAVE": 0, "SPZ_DEC_INTERLEAVE": 0}},
    "il_g1": {"enc": "interleaved, single tiles", "dec": "same", "defs": dict(_IL, SPZ_IL_GROUP=1)},
    "il_g4": {"enc": "interleaved, runs of 4 tiles", "dec": "same", "defs": dict(_IL, SPZ_IL_GROUP=4)},
    "il_g8": {"enc": "interleaved, runs of 8 tiles", "dec": "same", "defs": dict(_IL, SPZ_IL_GROUP=8)},
    "il_g16": {"enc": "interleaved, runs of 16 tiles", "dec": "same", "defs": dict(_IL, SPZ_IL_GROUP=16)},
    "il_g64": {"enc": "interleaved, runs of 64 tiles", "dec": "same", "defs": dict(_IL, SPZ_IL_GROUP=64)},
    "il_rot_g8": {"enc": "rotation tiles only, runs of 8", "dec": "same", "defs": dict(_IL, SPZ_IL_ONLY_ROT=1, SPZ_IL_GROUP=8)},
    "il_rot_g64": {"enc": "rotation tiles only, runs of 64", "dec": "same", "defs": dict(_IL, SPZ_IL_ONLY_ROT=1, SPZ_IL_GROUP=64)},
    "il_g256": {"enc": "interleaved, runs of 256 tiles", "dec": "same", "defs": dict(_IL, SPZ_IL_GROUP=256)},
    "il_g512": {"enc": "interleaved, runs of 512 tiles", "dec": "same", "defs": dict(_IL, SPZ_IL_GROUP=512)},
    "policy": {"enc": "shipped: interleave by policy, runs of 8", "dec": "same", "defs": {}},
    "policy_b": {"enc": "shipped, second copy (noise floor)", "dec": "same", "defs": {}},
    "mw1": {"enc": "policy, no minimum-waves bound (89 VGPRs, 5 waves per SIMD)", "dec": "policy", "defs": {"SPZ_ENC_MIN_WAVES": 1}},
    "mw7": {"enc": "policy, __launch_bounds__(256, 7)", "dec": "policy", "defs": {"SPZ_ENC_MIN_WAVES": 7}},
    "dec_wc": {"enc": "policy", "dec": "policy, a wave owns one contiguous 4 KiB span of its tile", "defs": {"SPZ_DEC_WC": 1}},
    "dec_u2": {"enc": "policy", "dec": "policy, 256 x 2 units (8 KiB of floats per block)", "defs": {"SPZ_DEC_UNROLL": 2}},
    "dec_u8": {"enc": "policy", "dec": "policy, 256 x 8 units (32 KiB of floats per block)", "defs": {"SPZ_DEC_UNROLL": 8}},
    "dec_b512": {"enc": "policy", "dec": "policy, 512 threads x 4 units", "defs": {"SPZ_DEC_BLOCK": 512}},
    "enc_nts": {"enc": "policy + non-temporal stores in encode", "dec": "policy", "defs": {"SPZ_ENC_NTS": 1}},
    "dec_ld": {"enc": "policy", "dec": "policy, ordinary (cached) loads", "defs": {"SPZ_DEC_NTL": 0}},
    "dec_st": {"enc": "policy", "dec": "policy, ordinary stores", "defs": {"SPZ_DEC_NTS": 0}},
    "u8_policy": {"enc": "256 x 8 units, policy", "dec": "256 x 8 units, policy", "defs": {"SPZ_ENC_UNROLL": 8, "SPZ_DEC_UNROLL": 8}},
}


def build(names):
    os.makedirs(VDIR, exist_ok=True)
    for name in names:
        defs = [f"-D{k}={v}" for k, v in VARIANTS[name]["defs"].items()]
        out = os.path.join(VDIR, f"libspz_amd_{name}.so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt", f"-I{ROOT}/include", "-shared",
               "-o", out] + [os.path.join(ROOT, "spz_amd", "csrc", f) for f in
                             ("spz_kernels.hip", "spz_abi.hip", "spz_hostpath.hip", "spz_ply_kernels.hip", "spz_median.hip", "spz_exchange.hip", "spz_lz77.hip", "spz_inflate_dev.hip", "spz_place.hip")] + ["-ldl"] + defs
        print(" ".join(cmd[-4:]), flush=True)
        subprocess.run(cmd, check=True)


def run(points, rounds, names, deg=3, version=3, batch=1):
    import statistics

    import torch

    from spz_amd import abi
    from spz_amd.synth import FIELDS, make_cloud_torch
    dev = torch.device("cuda:0")
    cloud = make_cloud_torch(points, deg, 3, dev)
    out = {k: torch.empty_like(cloud[k]) for k in FIELDS}
    lay = abi.stream_layout(points, deg, version)
    stream = torch.empty(lay.total_bytes, dtype=torch.uint8, device=dev)
    ref_stream = None
    pin = abi.CloudPtrs(*[cloud[k].data_ptr() for k in FIELDS])
    pout = abi.CloudPtrs(*[out[k].data_ptr() for k in FIELDS])
    hdr = abi.Header(version, points, deg, 12, 0, 0)
    libs = {}
    VARIANTS["prod"] = {"enc": "shipped defaults", "dec": "shipped defaults", "defs": {}}
    for name in names:
        path = abi.LIB_PATH if name == "prod" else os.path.join(VDIR, f"libspz_amd_{name}.so")
        if not os.path.exists(path):
            print(f"skip {name}: {path} missing")
            continue
        L = C.CDLL(path)
        abi.bind(L)
        libs[name] = L
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def enc(L):
        rc = L.spz_amd_encode_device(C.byref(pin), points, deg, 0, 6, version, stream.data_ptr(), stream.numel(), s)
        assert rc == 0, rc

    def dec(L):
        rc = L.spz_amd_decode_device(stream.data_ptr(), stream.numel(), C.byref(hdr), 6, C.byref(pout), s)
        assert rc == 0, rc

    times = {n: {"enc": [], "dec": [], "dec_cold": []} for n in libs}
    scrub = torch.empty(1 << 30, dtype=torch.uint8, device=dev)   # evicts the stream from the caches
    for name, L in libs.items():   # warm-up + cross-variant parity (outputs cleared first: a tile nobody wrote must show)
        stream.zero_()
        for k in FIELDS:
            out[k].fill_(float("nan"))
        enc(L); dec(L)
        torch.cuda.synchronize()
        if ref_stream is None:
            ref_stream = stream.clone()
            ref_out = {k: out[k].clone() for k in FIELDS}
        else:
            assert torch.equal(stream, ref_stream), f"{name}: stream differs from base"
            for k in FIELDS:
                assert torch.equal(out[k].view(torch.int32), ref_out[k].view(torch.int32)), f"{name}: {k} differs"
    for _ in range(rounds):
        for name, L in libs.items():
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            enc(L); dec(L)   # untimed: puts the caches in this pair's own steady state
            # batch > 1 (short kernels): `batch` launches back to back per timed region, so that the figure is the
            # device's rate and not the host's launch latency in front of a single 20 us kernel
            e[0].record()
            for _b in range(batch):
                enc(L)
            e[1].record()
            for _b in range(batch):
                dec(L)
            e[2].record()
            torch.cuda.synchronize()
            times[name]["enc"].append(e[0].elapsed_time(e[1]) / batch)
            times[name]["dec"].append(e[1].elapsed_time(e[2]) / batch)
            scrub.fill_(1)
            c = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            c[0].record(); dec(L); c[1].record()
            torch.cuda.synchronize()
            times[name]["dec_cold"].append(c[0].elapsed_time(c[1]))
    bpp = {0: 76, 1: 121, 2: 196, 3: 301}[deg] - (1 if version == 2 else 0)
    gb = points * bpp / 1e9
    rows = []
    for name in libs:
        r = {"variant": name, "points": points, "sh_degree": deg, "version": version, "launches_per_timed_region": batch,
             "enc_cfg": VARIANTS[name]["enc"], "dec_cfg": VARIANTS[name]["dec"]}
        for k in ("enc", "dec", "dec_cold"):
            med, mn = statistics.median(times[name][k]), min(times[name][k])
            r[f"{k}_ms_med"] = round(med, 4)
            r[f"{k}_ms_min"] = round(mn, 4)
            r[f"{k}_GBps_med"] = round(gb / (med * 1e-3), 1)
            r[f"{k}_frac_of_8TBps"] = round(gb / (med * 1e-3) / 8000.0, 3)
        r["pair_ms_med"] = round(r["enc_ms_med"] + r["dec_ms_med"], 4)
        rows.append(r)
        print(json.dumps(r), flush=True)
    return rows


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["build", "run"])
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--rounds", type=int, default=15)
    ap.add_argument("--deg", type=int, default=3)
    ap.add_argument("--version", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--variants", default=",".join(VARIANTS))
    a = ap.parse_args()
    names = [v for v in a.variants.split(",") if v]
    if a.mode == "build":
        build(names)
    else:
        run(a.points, a.rounds, names, a.deg, a.version, a.batch)
