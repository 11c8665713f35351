"""Device-resident encode / decode over the C ABI, with torch tensors as the memory owner.

torch is plumbing here: it allocates HBM and names the HIP stream; every byte of work is done
by libspz_amd.so through raw pointers (spz_amd.abi).  Clouds are dicts of flat float32 CUDA
tensors keyed like the reference's GaussianCloud fields (splat-types.h:101-115):
positions[3N], scales[3N], rotations[4N], alphas[N], colors[3N], sh[N*shDim*3].
"""
import ctypes as C

import torch

from . import abi
from .synth import FIELDS, SH_DIM, floats_per_point


def _stream_handle(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def _ptrs(cloud, sh_degree, n, device):
    p = abi.CloudPtrs()
    for k in FIELDS:
        t = cloud.get(k)
        need = n * floats_per_point(k, sh_degree)
        if need == 0:
            setattr(p, k, None)
            continue
        if t is None or t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.numel() != need:
            raise ValueError(f"cloud[{k!r}] must be a contiguous float32 CUDA tensor of {need} elements")
        if t.device != device:
            raise ValueError(f"cloud[{k!r}] is on {t.device}, expected {device}")
        setattr(p, k, t.data_ptr())
    return p


def alloc_cloud(n, sh_degree, device):
    return {k: torch.empty(n * floats_per_point(k, sh_degree), dtype=torch.float32, device=device) for k in FIELDS}


class _DeviceArray:
    """A typed view of device memory somebody else owns (kept alive through `owner`), for torch.as_tensor."""

    def __init__(self, ptr, count, typestr, owner):
        self.ptr, self.count, self.typestr, self.owner = int(ptr), int(count), typestr, owner

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.count,), "typestr": self.typestr, "data": (self.ptr, False), "version": 2, "strides": None}


class PlacedBuffers:
    """Device buffers of one resident cloud made by spz_amd_cloud_buffers_alloc (spz_place.hip): `cloud` is the usual
    dict of flat float32 CUDA tensors, `stream` a uint8 CUDA tensor (this object's, or the caller's), `report` what
    the placement probe saw.  The memory goes when the object does (or at free()); tensors made from it must not
    outlive it."""

    def __init__(self, raw, n, sh_degree, device, stream_tensor=None):
        self._raw, self._freed = raw, False
        self.cloud = {}
        for k in FIELDS:
            cnt = n * floats_per_point(k, sh_degree)
            ptr = getattr(raw.cloud, k)
            self.cloud[k] = (torch.as_tensor(_DeviceArray(ptr, cnt, "<f4", self), device=device) if cnt
                             else torch.empty(0, dtype=torch.float32, device=device))
        self.stream = stream_tensor if stream_tensor is not None else torch.as_tensor(
            _DeviceArray(raw.stream, raw.stream_capacity, "|u1", self), device=device)
        self.report = {"sh_placements_timed": int(raw.candidates), "probe_ms_first": round(float(raw.probe_ms_first), 4),
                       "probe_ms_chosen": round(float(raw.probe_ms_chosen), 4), "probe_ms_slowest": round(float(raw.probe_ms_worst), 4)}

    def free(self):
        if not self._freed:
            self._freed = True
            self.cloud, self.stream = {}, None
            abi.load_library().spz_amd_cloud_buffers_free(C.byref(self._raw))

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


def alloc_placed(n, sh_degree, device, version=3, stream_t=None, probe="decode", max_candidates=6, stream=None):
    """Buffers for a resident cloud of n points with the sh array placed for speed: the library times the
    launch they are for — probe "decode" (stream read, cloud written), "encode" (cloud read; the stream is OVERWRITTEN
    with zeros' code) or None — on up to `max_candidates` placements of the sh array and keeps the fastest.  `stream_t`:
    an existing stream tensor to place against (otherwise one is allocated with the small arrays)."""
    L = abi.load_library()
    raw = abi.CloudBuffers()
    mode = {None: 0, "none": 0, "decode": 1, "encode": 2}[probe]
    with torch.cuda.device(device):
        rc = L.spz_amd_cloud_buffers_alloc(int(n), int(sh_degree), int(version), stream_t.data_ptr() if stream_t is not None else None,
                                           mode, int(max_candidates), _stream_handle(stream), C.byref(raw))
    abi.check(rc, "spz_amd_cloud_buffers_alloc")
    return PlacedBuffers(raw, n, sh_degree, device, stream_t)


def make_header(num_points, sh_degree, version=3, fractional_bits=12, antialiased=False):
    return abi.Header(int(version), int(num_points), int(sh_degree), int(fractional_bits),
                      1 if antialiased else 0, 0)


def peek_header(stream_t, max_points=abi.REFERENCE_MAX_POINTS, stream=None):
    """Header checks of deserializePackedGaussians (load-spz.cc:551-568,591-594) on a device-resident
    stream.  Returns (status, Header or None)."""
    L = abi.load_library()
    h = abi.Header()
    with torch.cuda.device(stream_t.device):
        rc = L.spz_amd_peek_header_device(stream_t.data_ptr(), stream_t.numel(), int(max_points), C.byref(h),
                                          _stream_handle(stream))
    return rc, (h if rc == abi.OK else None)


class RawStream:
    """A device byte buffer known by address only (spz_amd_ipc_alloc / spz_amd_ipc_open).  `tensor()` gives a
    uint8 view of it for checks (through __cuda_array_interface__; the buffer must outlive the view)."""

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = int(ptr), int(nbytes)

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2, "strides": None}

    def tensor(self, device):
        return torch.as_tensor(self, device=device)


def encode(cloud, num_points, sh_degree, antialiased=False, from_coord=0, version=3, out=None, stream=None):
    """packGaussians + serializePackedGaussians on the GPU -> uint8 CUDA tensor holding the raw
    (pre-gzip) stream.  Asynchronous on `stream` (default: torch's current stream)."""
    L = abi.load_library()
    device = cloud["positions"].device if num_points else (out.device if out is not None else torch.device("cuda"))
    lay = abi.stream_layout(num_points, sh_degree, version)
    if out is None:
        out = torch.empty(lay.total_bytes, dtype=torch.uint8, device=device)
    p = _ptrs(cloud, sh_degree, num_points, device)
    with torch.cuda.device(device):
        rc = L.spz_amd_encode_device(C.byref(p), num_points, sh_degree, int(bool(antialiased)), from_coord, version,
                                     out.data_ptr(), out.numel(), _stream_handle(stream))
    abi.check(rc, "spz_amd_encode_device")
    return out[:lay.total_bytes]


def decode(stream_t, header, to_coord=0, out=None, stream=None):
    """unpackGaussians (+ fused coordinate flip) on the GPU.  `header` is an abi.Header (from
    abi.peek_header on host bytes, or make_header for a stream this process encoded)."""
    L = abi.load_library()
    n, deg = header.num_points, header.sh_degree
    if out is None:
        out = alloc_cloud(n, deg, stream_t.device)
    p = _ptrs(out, deg, n, stream_t.device)
    with torch.cuda.device(stream_t.device):
        rc = L.spz_amd_decode_device(stream_t.data_ptr(), stream_t.numel(), C.byref(header), to_coord, C.byref(p),
                                     _stream_handle(stream))
    abi.check(rc, "spz_amd_decode_device")
    return out


def encode_shard(cloud, first, count, num_points_total, sh_degree, out, antialiased=False, from_coord=0, version=3,
                 write_header=False, stream=None, section_mask=abi.ALL_SECTIONS):
    """Encode points [first, first+count) (cloud holds only those) into the FULL stream `out`: a uint8 CUDA
    tensor, or a RawStream (pointer + size: e.g. another process's buffer mapped over IPC).  section_mask
    restricts the launch to some of the six sections (abi.SMALL_SECTIONS / abi.SH_SECTION)."""
    L = abi.load_library()
    device = cloud["positions"].device
    p = _ptrs(cloud, sh_degree, count, device)
    ptr, size = (out.ptr, out.nbytes) if isinstance(out, RawStream) else (out.data_ptr(), out.numel())
    with torch.cuda.device(device):
        rc = L.spz_amd_encode_shard_sections_device(C.byref(p), first, count, num_points_total, sh_degree,
                                                    int(bool(antialiased)), from_coord, version, int(bool(write_header)),
                                                    int(section_mask), ptr, size, _stream_handle(stream))
    abi.check(rc, "spz_amd_encode_shard_sections_device")
    return out


def decode_shard(stream_t, header, first, count, to_coord=0, out=None, stream=None):
    L = abi.load_library()
    deg = header.sh_degree
    if out is None:
        out = alloc_cloud(count, deg, stream_t.device)
    p = _ptrs(out, deg, count, stream_t.device)
    with torch.cuda.device(stream_t.device):
        rc = L.spz_amd_decode_shard_device(stream_t.data_ptr(), stream_t.numel(), C.byref(header), first, count,
                                           to_coord, C.byref(p), _stream_handle(stream))
    abi.check(rc, "spz_amd_decode_shard_device")
    return out


def decode_gather(stream_t, header, indices, to_coord=0, out=None, stream=None):
    """Decode only the points `indices` (uint32/int32 CUDA tensor) of a packed device stream."""
    L = abi.load_library()
    count, deg = indices.numel(), header.sh_degree
    if indices.dtype not in (torch.int32, torch.uint32) or not indices.is_cuda or not indices.is_contiguous():
        raise ValueError("indices must be a contiguous int32/uint32 CUDA tensor")
    if out is None:
        out = alloc_cloud(count, deg, stream_t.device)
    p = _ptrs(out, deg, count, stream_t.device)
    with torch.cuda.device(stream_t.device):
        rc = L.spz_amd_decode_gather_device(stream_t.data_ptr(), stream_t.numel(), C.byref(header), indices.data_ptr(),
                                            count, to_coord, C.byref(p), _stream_handle(stream))
    abi.check(rc, "spz_amd_decode_gather_device")
    return out


def convert_coordinates(cloud, num_points, sh_degree, from_coord, to_coord, stream=None):
    """In-place GaussianCloud::convertCoordinates on device tensors (positions, rotations, sh)."""
    L = abi.load_library()
    dev = cloud["positions"].device

    def ptr(k):
        t = cloud.get(k)
        return t.data_ptr() if t is not None and t.numel() else None

    with torch.cuda.device(dev):
        rc = L.spz_amd_convert_coordinates_device(ptr("positions"), ptr("rotations"), ptr("sh"), num_points,
                                                  sh_degree, from_coord, to_coord, _stream_handle(stream))
    abi.check(rc, "spz_amd_convert_coordinates_device")
    return cloud


def to_device(cloud_np, device):
    return {k: torch.from_numpy(cloud_np[k]).to(device) for k in FIELDS}


def to_numpy(cloud_t):
    return {k: cloud_t[k].cpu().numpy() for k in FIELDS}


__all__ = ["encode", "decode", "encode_shard", "decode_shard", "decode_gather", "peek_header", "convert_coordinates",
           "alloc_cloud",
           "make_header", "to_device", "to_numpy", "SH_DIM"]
