#!/usr/bin/env python3
"""Randomized campaign for the byte-identical multi-threaded gzip writer (spz_amd/csrc/spz_deflate.cpp): for every
input, exactgz::compress (through spz._compress_gzipped_exact, no prefix self-check, random thread count and job
size) must return exactly what zlib 1.2.11 returns with the reference's parameters (load-spz.cc:190), or decline.

  make gzip-campaign            # ~9 300 inputs, host only (no GPU): textures, mixtures, real-stream-shaped
  python tools/gzip_campaign.py --inputs 500 --seed 7 --max-mib 6

Inputs: the five textures of tests/test_exact_gzip.py plus long-distance copies, section-shaped mixtures
(24-bit fixed point | bytes | bucketed sh like a raw .spz stream of every degree) and random splices of all of
them; sizes from 128 KiB (the writer's minimum) to --max-mib; every chunking (4..32 windows per job) and 2..16
threads.  Prints one JSON line; exit code 1 on the first difference (the input's recipe is printed)."""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def zlib_gzip(b):
    co = zlib.compressobj(-1, zlib.DEFLATED, 16 + 15, 9, zlib.Z_DEFAULT_STRATEGY)
    return co.compress(b) + co.flush()


def stream_like(n_points, deg, rng):
    """The byte statistics of a raw .spz stream (header | 24-bit positions | alphas | colours | scales | rotations | sh)."""
    d = {0: 0, 1: 9, 2: 24, 3: 45}[deg]
    pos = (rng.uniform(-10, 10, 3 * n_points) * 4096).astype(np.int32)
    pos3 = np.stack([pos & 255, (pos >> 8) & 255, (pos >> 16) & 255], axis=1).astype(np.uint8).reshape(-1)
    sig = lambda x: 1 / (1 + np.exp(-x))
    parts = [np.frombuffer(b"NGSP\x03\0\0\0" + int(n_points).to_bytes(4, "little") + bytes([deg, 12, 0, 0]), np.uint8), pos3,
             np.clip(np.round(sig(rng.normal(0, 3, n_points)) * 255), 0, 255).astype(np.uint8),
             np.clip(np.round(rng.normal(0, 1, 3 * n_points) * 38.25 + 127.5), 0, 255).astype(np.uint8),
             np.clip(np.round((rng.uniform(-8, 0, 3 * n_points) + 10) * 16), 0, 255).astype(np.uint8),
             rng.integers(0, 256, 4 * n_points, dtype=np.uint8),
             np.clip((np.round(rng.normal(0, 0.25, d * n_points) * 128 / 16) * 16 + 128), 0, 255).astype(np.uint8)]
    return np.concatenate(parts).tobytes()


def make_input(i, rng, max_bytes):
    from test_exact_gzip import make
    n = int(rng.integers(128 << 10, max_bytes))
    kind = ("nibbles", "bytes", "words", "runs", "sh_like", "copies", "stream", "splice")[i % 8]
    if kind in ("nibbles", "bytes", "words", "runs", "sh_like"):
        return kind, make(kind, n, rng)
    if kind == "copies":   # long-distance repeats: the same 1..40 KiB pieces come back at distances around the 32 KiB window
        pieces = [rng.integers(0, 256, int(rng.integers(1 << 10, 40 << 10)), dtype=np.uint8).tobytes() for _ in range(6)]
        out = bytearray()
        while len(out) < n:
            out += pieces[int(rng.integers(0, 6))]
            out += rng.integers(0, 64, int(rng.integers(0, 3000)), dtype=np.uint8).tobytes()
        return kind, bytes(out[:n])
    if kind == "stream":
        deg = int(rng.integers(0, 4))
        per = 20 + {0: 0, 1: 9, 2: 24, 3: 45}[deg]
        return f"stream sh{deg}", stream_like(max(1, n // per), deg, rng)
    parts, total = [], 0   # splice of other kinds at arbitrary cut points
    while total < n:
        k, b = make_input(int(rng.integers(0, 7)), rng, max(256 << 10, max_bytes // 3))
        cut = int(rng.integers(1, len(b)))
        parts.append(b[:cut])
        total += cut
    return "splice", b"".join(parts)[:n]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inputs", type=int, default=9300)
    ap.add_argument("--seed", type=int, default=20261004)
    ap.add_argument("--max-mib", type=float, default=3.0)
    ap.add_argument("--writer", choices=("host", "model", "device"), default="host",
                    help="host: the multi-threaded parse of spz_deflate.cpp; model: the serial host model of the device's "
                         "stages (spz_lz77_model.cpp; inputs of 512 KiB and more); device: compressGzipped with "
                         "SPZ_AMD_GZIP_DEVICE=1 (needs a GPU; a member counts only if the device did the parse)")
    a = ap.parse_args()
    if zlib.ZLIB_RUNTIME_VERSION != "1.2.11":
        print(json.dumps({"skipped": f"zlib {zlib.ZLIB_RUNTIME_VERSION}: the writer restates 1.2.11 and stands down"}))
        return 0
    import spz_amd.spz as spz
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    same = declined = total_bytes = 0
    kinds = {}
    declined_kinds = {}
    device_inflates = 0
    for i in range(a.inputs):
        kind, data = make_input(i, rng, int(a.max_mib * (1 << 20)))
        threads = int(rng.integers(2, 17))
        windows = int(rng.integers(4, 33))
        if a.writer == "model":
            if len(data) < (512 << 10):
                data = (data * (1 + (512 << 10) // len(data)))[: (512 << 10) + len(data) % 4099]
            got = spz._compress_gzipped_exact_model(data, threads, 0)
        elif a.writer == "device":
            if len(data) < (1 << 20):
                data = (data * (1 + (1 << 20) // len(data)))[: (1 << 20) + len(data) % 4099]
            os.environ["SPZ_AMD_GZIP_DEVICE"] = "1"
            before = spz._device_gzip_parse_count()
            got = spz._compress_gzipped(data)
            if spz._device_gzip_parse_count() == before:
                got = None if got == zlib_gzip(data) else got   # the device declined; the fallback's bytes must still be zlib's
            # and back: the member through the device reader (or, where it declines, the host readers) must give the input
            os.environ["SPZ_AMD_GUNZIP_DEVICE"] = "1"
            inflates_before = spz._device_inflate_count()
            if spz._decompress_gzipped(got if got is not None else zlib_gzip(data)) != data:
                print(json.dumps({"FAILED": "inflate", "input": i, "kind": kind, "bytes": len(data), "seed": a.seed}))
                return 1
            device_inflates += spz._device_inflate_count() - inflates_before
        else:
            got = spz._compress_gzipped_exact(data, threads, windows, 0)
        kinds[kind.split()[0]] = kinds.get(kind.split()[0], 0) + 1
        total_bytes += len(data)
        if got is None:
            declined += 1
            declined_kinds[kind.split()[0]] = declined_kinds.get(kind.split()[0], 0) + 1
            continue
        if got != zlib_gzip(data):
            print(json.dumps({"FAILED": True, "input": i, "kind": kind, "bytes": len(data), "threads": threads,
                              "windows_per_job": windows, "seed": a.seed}))
            return 1
        same += 1
        if (i + 1) % 500 == 0:
            print(f"[gzip-campaign] {i + 1}/{a.inputs} inputs, {same} identical, {declined} declined, {time.time() - t0:.0f} s",
                  file=sys.stderr, flush=True)
    print(json.dumps({"writer": a.writer, "inputs": a.inputs, "identical_to_zlib": same, "declined": declined, "different": 0, "seed": a.seed,
                      "total_MB": round(total_bytes / 1e6, 1), "by_kind": kinds, "declined_by_kind": declined_kinds,
                      **({"inflated_back_identical": a.inputs, "of_them_on_the_device": device_inflates} if a.writer == "device" else {}), "seconds": round(time.time() - t0, 1)}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
