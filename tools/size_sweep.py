#!/usr/bin/env python3
"""saveSpz / loadSpz wall-clock from 60 k to 10 M points with every container route forced on and off (VERDICT r02
next #5): are the route thresholds (device gzip >= 8 MiB, exact host writer >= 1 MiB, parallel inflate >= 4 MiB, device
inflate >= 8 MiB) where the curves cross?  Drives spz_amd/bin/host_bench (C++ boundary, no binding in the way); first
and steady-state figures per setting.

  python tools/size_sweep.py > profiles/r03_size_sweep.json
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "spz_amd", "bin", "host_bench")

# (points, sh degree): cfg1 = 60 k SH3, 250 k SH3, cfg2 = 1 M SH0, 1 M SH3 (65 MB), 2 M SH3, 4 M SH3, cfg3 = 10 M SH3
WORKLOADS = [(60_000, 3), (250_000, 3), (1_000_000, 0), (500_000, 3), (1_000_000, 3), (2_000_000, 3), (4_000_000, 3), (10_000_000, 3)]
ROUTES = {
    "default": {},
    "gzip_device": {"SPZ_AMD_GZIP_DEVICE": "1", "SPZ_AMD_GUNZIP_DEVICE": "1"},
    "gzip_host_exact": {"SPZ_AMD_GZIP_DEVICE": "0", "SPZ_AMD_GUNZIP_DEVICE": "0"},
    "zlib_only": {"SPZ_AMD_GZIP_DEVICE": "0", "SPZ_AMD_GUNZIP_DEVICE": "0", "SPZ_AMD_GZIP_EXACT_THREADS": "1", "SPZ_AMD_GUNZIP_THREADS": "1",
                  "SPZ_AMD_NO_LIBDEFLATE": "1"},
}


def main():
    rows = []
    for n, deg in WORKLOADS:
        stream_mb = (16 + n * (20 + {0: 0, 1: 9, 2: 24, 3: 45}[deg])) / 1e6
        for route, env in ROUTES.items():
            if route == "zlib_only" and stream_mb > 140:
                continue   # zlib itself at 11 MB/s: minutes per file, and known
            e = dict(os.environ)
            e.update(env)
            reps = 3 if stream_mb < 200 else 2
            p = subprocess.run([BENCH, str(n), str(deg), str(reps), "1"], env=e, capture_output=True, text=True, timeout=900)
            if p.returncode != 0:
                rows.append({"points": n, "sh_degree": deg, "route": route, "error": p.stderr[-300:]})
                continue
            d = json.loads(p.stdout.strip().splitlines()[-1])
            rows.append({"points": n, "sh_degree": deg, "stream_MB": round(stream_mb, 1), "route": route,
                         "save_spz_first_s": d.get("save_spz_first_s"), "save_spz_s": d.get("save_spz_s"),
                         "load_spz_first_s": d.get("load_spz_first_s"), "load_spz_s": d.get("load_spz_s"),
                         "spz_bytes": d.get("spz_bytes"), "load_ok": d.get("load_ok"),
                         "pack_s": d.get("pack_to_stream_fresh_vector_s"), "unpack_s": d.get("unpack_from_stream_s")})
            print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    print(json.dumps({"host_cores_note": "GPU box: 16 usable CPUs per GPU (cgroup quota)", "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
