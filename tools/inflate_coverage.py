#!/usr/bin/env python3
"""What does the device reader (spz_inflate_dev.hip) take, and why does it decline the rest?

The reference's reader accepts any gzip member (decompressGzipped, load-spz.cc:141-184); .spz files come from other
writers too (other zlib levels, memLevel 8, libdeflate- or miniz-based ports).  A decline costs only time (the host
readers give the same bytes), so it never shows in a test — this campaign counts, per writer setting, how many members
the device inflated and the reason for every one it did not (spz_amd_inflate_last_decline).

  python tools/inflate_coverage.py [--per-cell 6] > profiles/r03_inflate_coverage.json
"""
import argparse
import json
import os
import sys
import zlib
from collections import Counter, defaultdict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def textures(rng, n):
    from test_exact_gzip import make
    yield "sh_like", make("sh_like", n, rng)
    yield "words", make("words", n, rng)
    yield "nibbles", make("nibbles", n, rng)
    yield "runs", make("runs", n, rng)
    # incompressible stretches between compressible ones: runs of stored blocks inside the member
    parts = []
    left = n
    while left > 0:
        k = int(min(left, rng.integers(100_000, 3_000_000)))
        parts.append(rng.integers(0, 256, k, dtype=np.uint8).tobytes() if rng.integers(0, 2) else make("sh_like", k, rng))
        left -= k
    yield "stored_mix", b"".join(parts)


def spz_stream(n_points, deg, seed):
    import spz_amd.spz as spz
    from spz_amd.synth import FIELDS, make_cloud_numpy
    c = make_cloud_numpy(n_points, deg, seed)
    g = spz.GaussianCloud()
    g.sh_degree = deg
    for k in FIELDS:
        if len(c[k]):
            setattr(g, k, c[k])
    return spz._pack_to_stream(g, spz.PackOptions())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--per-cell", type=int, default=4)
    a = ap.parse_args()
    os.environ["SPZ_AMD_GUNZIP_DEVICE"] = "1"
    import spz_amd.spz as spz
    rng = np.random.default_rng(2025)
    writers = [("zlib-1", 1, 9), ("zlib-6", 6, 9), ("zlib-9", 9, 9), ("zlib-6-mem8", 6, 8), ("zlib-1-mem8", 1, 8)]
    cells = defaultdict(Counter)
    detail = []
    total = 0
    for rep in range(a.per_cell):
        sizes = [int(rng.integers(2_000_000, 6_000_000)), int(rng.integers(10_000_000, 30_000_000))]
        inputs = []
        for n in sizes:
            inputs.extend((f"{name}", data) for name, data in textures(rng, n))
        inputs.append(("spz_stream_sh3", spz_stream(int(rng.integers(150_000, 500_000)), 3, 100 + rep)))
        inputs.append(("spz_stream_sh0", spz_stream(int(rng.integers(400_000, 1_500_000)), 0, 200 + rep)))
        for name, data in inputs:
            for wname, level, mem in writers:
                co = zlib.compressobj(level, zlib.DEFLATED, 16 + 15, mem, zlib.Z_DEFAULT_STRATEGY)
                member = co.compress(data) + co.flush()
                before = spz._device_inflate_count()
                out = spz._decompress_gzipped(member)
                if out != data:
                    print(json.dumps({"error": "wrong bytes", "texture": name, "writer": wname, "n": len(data)}))
                    sys.exit(1)
                took = spz._device_inflate_count() == before + 1
                reason = "device" if took else (spz._device_inflate_last_decline() or "not-asked")
                cells[wname][reason] += 1
                detail.append({"texture": name, "writer": wname, "bytes": len(data), "member_bytes": len(member), "result": reason})
                total += 1
    by_texture = defaultdict(Counter)
    for d in detail:
        by_texture[d["texture"]][d["result"]] += 1
    print(json.dumps({"members": total, "all_bytes_right": True,
                      "by_writer": {w: dict(c) for w, c in cells.items()},
                      "by_texture": {t: dict(c) for t, c in by_texture.items()},
                      "declined": [d for d in detail if d["result"] != "device"]}, indent=1))


if __name__ == "__main__":
    main()
