"""spz_amd — MI355X-native SPZ (Gaussian-splat) pack/unpack hot path.

Layout:
  csrc/spz_kernels.hip  hand-written HIP pack/unpack/flip kernels (gfx950)
  csrc/spz_ply_kernels.hip  .ply row <-> cloud shuffles
  csrc/spz_median.hip   radix selection of the median scale sum (medianVolume)
  csrc/spz_abi.hip      the C ABI (include/spz_amd.h) over them: device-pointer entry points
  csrc/spz_hostpath.hip host-pointer entry points: device workspace + chunked H2D/kernel/D2H pipeline
  csrc/spz_exchange.hip multi-GPU exchange: native RCCL gatherv/scatterv, IPC-mapped root stream
  csrc/spz_lz77.hip     the gzip writer's stages on the device: zlib's level-6 parse (exact) and Huffman bit packing
  csrc/spz_lz77_core.hpp  ... their stage functions, shared with the serial host model (spz_lz77_model.cpp, tests)
  csrc/spz_huff_core.hpp  ... zlib's tree construction for a deflate block, shared by the host writer and the tree kernel
  csrc/spz_inflate_dev.hip  inflate of single-stream gzip members on the device (CRC-verified)
  csrc/spz_inflate_core.hpp ... the deflate decoder shared by the host and the device reader
  csrc/spz_host.cpp     C++ drop-in layer: namespace spz saveSpz/loadSpz/... + host gzip
  csrc/spz_deflate.cpp  multi-threaded gzip writer with zlib's exact bytes (the default container stage)
  csrc/spz_inflate.cpp  multi-threaded, CRC-verified inflate of ordinary single-stream members
  csrc/spz_py.cpp       Python module `spz_amd.spz` with the reference nanobind shim's surface
  abi.py                ctypes binding of the C ABI
  device.py             device-resident encode/decode on torch-owned HBM
  shard.py              point-range sharding across GPUs + gatherv of the byte stream
  synth.py              seeded synthetic clouds

Nothing here falls back to the CPU: without libspz_amd.so or a HIP device, calls raise.
"""
__version__ = "0.1.0"


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels carry their own libamdhip64.so (SONAME
    libamdhip64.so.7) and look it up by FILE name, libspz_amd.so asks for the SONAME: loaded after
    torch it binds to torch's copy, loaded BEFORE torch the process would end up with two runtimes,
    and the second one to initialise finds no device (hipErrorNoDevice).  So, when a torch with a
    bundled runtime is installed, that copy is mapped first — without importing torch; either import
    order then works.  Without torch the system runtime in libspz_amd.so's RUNPATH is used."""
    import ctypes
    import importlib.util
    import os
    import sys
    if "torch" in sys.modules or os.environ.get("SPZ_AMD_NO_HIP_PRELOAD"):   # host-only helper processes
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


_share_hip_runtime_with_torch()
