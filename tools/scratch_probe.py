#!/usr/bin/env python3
"""Does the container stage return its scratch when SPZ_AMD_SCRATCH_KEEP_MIB=0?  Free device memory around calls,
for the writer's three ways of using the device and for the reader."""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
os.environ["SPZ_AMD_GZIP_DEVICE"] = "1"
os.environ["SPZ_AMD_GUNZIP_DEVICE"] = "1"
os.environ["SPZ_AMD_SCRATCH_KEEP_MIB"] = "0"
import spz_amd.spz as spz
from spz_amd import abi
from test_exact_gzip import make
def free_mib():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] >> 20
data = make("sh_like", 40_000_000, np.random.default_rng(1))
print("start", free_mib(), flush=True)
member = None
for name, env in (("trees on device", {}), ("trees on host", {"SPZ_AMD_GZIP_DEVICE_TREES": "0"}), ("parse only", {"SPZ_AMD_GZIP_DEVICE_HUFFMAN": "0"})):
    for k, v in env.items():
        os.environ[k] = v
    for k in range(2):
        member = spz._compress_gzipped(data)
        print(f"writer, {name}: after call {k}: {free_mib()}", flush=True)
    for k in env:
        os.environ.pop(k)
for k in range(2):
    assert spz._decompress_gzipped(member) == data
    print(f"reader: after call {k}: {free_mib()}", flush=True)
abi.load_library().spz_amd_release_device_memory()
print("after release", free_mib(), flush=True)
