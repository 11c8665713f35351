#!/usr/bin/env python3
"""loadSplatFromPly / saveSplatToPly of a large cloud at the Python boundary (a file in /dev/shm), best of 3.
  SPZ_AMD_FILE_IO_THREADS=0 python tools/ply_file_bench.py [points]   # the reference's single stream read"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import spz_amd.spz as spz
    from spz_amd.synth import FIELDS, make_cloud_numpy
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    c = make_cloud_numpy(n, 3, 3)
    g = spz.GaussianCloud()
    g.sh_degree = 3
    for k in FIELDS:
        setattr(g, k, c[k])
    path = f"/dev/shm/spz_ply_bench_{os.getpid()}.ply"
    o, u = spz.PackOptions(), spz.UnpackOptions()
    t_save = t_load = 1e9
    try:
        for _ in range(3):
            t0 = time.perf_counter()
            assert spz.save_splat_to_ply(g, o, path)
            t_save = min(t_save, time.perf_counter() - t0)
        for _ in range(3):
            t0 = time.perf_counter()
            d = spz.load_splat_from_ply(path, u)
            t_load = min(t_load, time.perf_counter() - t0)
            assert d.num_points == n
        size = os.path.getsize(path)
    finally:
        if os.path.exists(path):
            os.remove(path)
    print(json.dumps({"points": n, "ply_MB": round(size / 1e6, 1), "file_io_threads": os.environ.get("SPZ_AMD_FILE_IO_THREADS", "default"),
                      "save_splat_to_ply_s": round(t_save, 3), "load_splat_from_ply_s": round(t_load, 3)}))


if __name__ == "__main__":
    main()
