"""spz_amd — MI355X-native SPZ (Gaussian-splat) pack/unpack hot path.

Layout:
  csrc/spz_kernels.hip  hand-written HIP pack/unpack/flip kernels (gfx950)
  csrc/spz_ply_kernels.hip  .ply row <-> cloud shuffles
  csrc/spz_abi.hip      the C ABI (include/spz_amd.h) over them
  csrc/spz_host.cpp     C++ drop-in layer: namespace spz saveSpz/loadSpz/... + host gzip
  csrc/spz_py.cpp       Python module `spz_amd.spz` with the reference nanobind shim's surface
  abi.py                ctypes binding of the C ABI
  device.py             device-resident encode/decode on torch-owned HBM
  shard.py              point-range sharding across GPUs + gatherv of the byte stream
  synth.py              seeded synthetic clouds

Nothing here falls back to the CPU: without libspz_amd.so or a HIP device, calls raise.
"""
__version__ = "0.1.0"
