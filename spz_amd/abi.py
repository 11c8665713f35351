"""ctypes binding of the C ABI (include/spz_amd.h, spz_amd/lib/libspz_amd.so).

This is the same boundary a cgo / JNI / N-API binding would use: plain pointers and sizes.
PyTorch only supplies device memory and streams to it (``tensor.data_ptr()``,
``torch.cuda.current_stream().cuda_stream``).  There is no CPU fallback: if the shared
library is missing or no HIP device is usable, calls raise.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SPZ_AMD_LIB: another build of the library (tools/tune.py variants under a profiler); default: the in-tree build
LIB_PATH = os.environ.get("SPZ_AMD_LIB") or os.path.join(HERE, "lib", "libspz_amd.so")

OK = 0
ERR_INVALID_ARG = -1
ERR_HEADER_NOT_FOUND = -2
ERR_VERSION = -3
ERR_TOO_MANY_POINTS = -4
ERR_SH_DEGREE = -5
ERR_SHORT_STREAM = -6
ERR_CAPACITY = -7
ERR_NO_DEVICE = -8
ERR_HIP = -9
ERR_UNSUPPORTED = -10
ERR_COMM = -11
ERR_VERIFY = -12

UNSPECIFIED, LDB, RDB, LUB, RUB, LDF, RDF, LUF, RUF = range(9)
NUM_SECTIONS = 6
SEC_POSITIONS, SEC_ALPHAS, SEC_COLORS, SEC_SCALES, SEC_ROTATIONS, SEC_SH = range(6)
REFERENCE_MAX_POINTS = 10_000_000

# Every symbol include/spz_amd.h declares (tests check the library exports all of them).
EXPORTS = (
    "spz_amd_abi_version", "spz_amd_status_string", "spz_amd_device_count", "spz_amd_last_hip_error",
    "spz_amd_release_device_memory",
    "spz_amd_stream_layout", "spz_amd_write_header", "spz_amd_peek_header", "spz_amd_peek_header_ex",
    "spz_amd_peek_header_device",
    "spz_amd_encode_device", "spz_amd_decode_device", "spz_amd_encode_shard_device",
    "spz_amd_decode_shard_device", "spz_amd_decode_gather_device", "spz_amd_decode_gather_host",
    "spz_amd_convert_coordinates_device",
    "spz_amd_encode_host",
    "spz_amd_decode_host", "spz_amd_decode_host_ex", "spz_amd_decode_host_from_device", "spz_amd_convert_coordinates_host", "spz_amd_get_tables",
    "spz_amd_ply_default_columns", "spz_amd_ply_rows_to_cloud_device", "spz_amd_cloud_to_ply_rows_device",
    "spz_amd_ply_rows_to_cloud_host", "spz_amd_cloud_to_ply_rows_host",
    "spz_amd_median_scale_sum_device", "spz_amd_median_scale_sum_host",
    "spz_amd_selftest_device",
    "spz_amd_encode_shard_sections_device", "spz_amd_shard_fragments",
    "spz_amd_rccl_available", "spz_amd_last_rccl_error", "spz_amd_rccl_unique_id", "spz_amd_rccl_comm_init",
    "spz_amd_rccl_comm_destroy", "spz_amd_gatherv_rccl", "spz_amd_scatterv_rccl",
    "spz_amd_ipc_alloc", "spz_amd_ipc_free", "spz_amd_ipc_open", "spz_amd_ipc_close",
    "spz_amd_zlib_parse_open", "spz_amd_zlib_parse_open_ex", "spz_amd_zlib_parse_fetch", "spz_amd_zlib_parse_close",
    "spz_amd_zlib_parse_append", "spz_amd_zlib_block_stats", "spz_amd_zlib_encode_blocks", "spz_amd_zlib_verify_member",
    "spz_amd_zlib_encode_group", "spz_amd_zlib_encode_finish", "spz_amd_zlib_parse_open_dev", "spz_amd_encode_host_keep", "spz_amd_kept_stream_release",
    "spz_amd_zlib_block_trees", "spz_amd_zlib_encode_planned", "spz_amd_zlib_encode_finish_ex",
    "spz_amd_inflate_open", "spz_amd_inflate_open_ex", "spz_amd_inflate_open_device", "spz_amd_inflate_equals_device", "spz_amd_inflate_crc_piece_bytes", "spz_amd_inflate_piece_crcs", "spz_amd_inflate_fetch",
    "spz_amd_inflate_device_data", "spz_amd_inflate_close", "spz_amd_stream_to_device", "spz_amd_inflate_last_decline",
    "spz_amd_cloud_buffers_alloc", "spz_amd_cloud_buffers_free",
    "spz_amd_zlib_session_open", "spz_amd_zlib_session_feed", "spz_amd_zlib_session_close", "spz_amd_zlib_parse_open_session",
    "spz_amd_encode_host_keep_session", "spz_amd_encode_host_keep_session_tail", "spz_amd_decode_gather_host_from_device",
)

RCCL_UNIQUE_ID_BYTES = 128
IPC_HANDLE_BYTES = 64
ALL_SECTIONS = 0x3f
SMALL_SECTIONS = 0x1f      # positions, alphas, colors, scales, rotations (20 B/point for v3)
SH_SECTION = 0x20


class Header(C.Structure):
    _fields_ = [("version", C.c_uint32), ("num_points", C.c_uint32), ("sh_degree", C.c_uint8),
                ("fractional_bits", C.c_uint8), ("flags", C.c_uint8), ("reserved", C.c_uint8)]

    @property
    def antialiased(self):
        return bool(self.flags & 1)


class Layout(C.Structure):
    _fields_ = [("total_bytes", C.c_uint64), ("offset", C.c_uint64 * NUM_SECTIONS),
                ("bytes", C.c_uint64 * NUM_SECTIONS), ("bytes_per_point", C.c_uint32 * NUM_SECTIONS)]


class CloudPtrs(C.Structure):
    """spz_amd_cloud_in / spz_amd_cloud_out (same layout: six pointers)."""
    _fields_ = [(k, C.c_void_p) for k in ("positions", "scales", "rotations", "alphas", "colors", "sh")]


class CloudBuffers(C.Structure):
    """spz_amd_cloud_buffers: device buffers made (and placed) by spz_amd_cloud_buffers_alloc."""
    _fields_ = [("cloud", CloudPtrs), ("stream", C.c_void_p), ("stream_capacity", C.c_size_t), ("owner", C.c_void_p),
                ("candidates", C.c_int32), ("probe_ms_first", C.c_float), ("probe_ms_chosen", C.c_float),
                ("probe_ms_worst", C.c_float)]


class PlyColumns(C.Structure):
    """spz_amd_ply_columns: column map of a .ply vertex row."""
    _fields_ = [("stride", C.c_int32), ("sh_dim", C.c_int32), ("position", C.c_int32 * 3), ("scale", C.c_int32 * 3),
                ("rotation", C.c_int32 * 4), ("alpha", C.c_int32), ("color", C.c_int32 * 3), ("sh", C.c_int32 * 45)]


class Fragments(C.Structure):
    """spz_amd_fragments: the six byte ranges of a point-range shard."""
    _fields_ = [("global_offset", C.c_uint64 * NUM_SECTIONS), ("local_offset", C.c_uint64 * NUM_SECTIONS),
                ("bytes", C.c_uint64 * NUM_SECTIONS)]


class SpzAmdError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        super().__init__(f"{where}: {status_string(status)} (status {status})")


MEDIAN_WORKSPACE_BYTES = 8192   # SPZ_AMD_MEDIAN_WORKSPACE_BYTES

_lib = None


def load_library():
    """Load libspz_amd.so.  torch is imported first (when available) so that the HIP runtime
    torch ships is the one both share (same SONAME libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `make` (or __graft_entry__.build()); "
                           "spz_amd has no CPU fallback")
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    _lib = bind(C.CDLL(LIB_PATH))
    return _lib


def bind(L):
    """Declare the C ABI's argument / result types on a loaded library handle."""
    vp, u64, i32, sz = C.c_void_p, C.c_uint64, C.c_int, C.c_size_t
    L.spz_amd_abi_version.restype = i32
    L.spz_amd_status_string.restype = C.c_char_p
    L.spz_amd_status_string.argtypes = [i32]
    L.spz_amd_device_count.restype = i32
    L.spz_amd_last_hip_error.restype = i32
    L.spz_amd_release_device_memory.restype = i32
    L.spz_amd_stream_layout.restype = i32
    L.spz_amd_stream_layout.argtypes = [u64, i32, i32, C.POINTER(Layout)]
    L.spz_amd_write_header.restype = i32
    L.spz_amd_write_header.argtypes = [C.POINTER(Header), vp]
    L.spz_amd_peek_header.restype = i32
    L.spz_amd_peek_header.argtypes = [vp, sz, C.POINTER(Header)]
    L.spz_amd_peek_header_ex.restype = i32
    L.spz_amd_peek_header_ex.argtypes = [vp, sz, u64, C.POINTER(Header)]
    L.spz_amd_peek_header_device.restype = i32
    L.spz_amd_peek_header_device.argtypes = [vp, sz, u64, C.POINTER(Header), vp]
    L.spz_amd_encode_device.restype = i32
    L.spz_amd_encode_device.argtypes = [C.POINTER(CloudPtrs), u64, i32, i32, i32, i32, vp, sz, vp]
    L.spz_amd_decode_device.restype = i32
    L.spz_amd_decode_device.argtypes = [vp, sz, C.POINTER(Header), i32, C.POINTER(CloudPtrs), vp]
    L.spz_amd_encode_shard_device.restype = i32
    L.spz_amd_encode_shard_device.argtypes = [C.POINTER(CloudPtrs), u64, u64, u64, i32, i32, i32, i32, i32, vp, sz, vp]
    L.spz_amd_decode_shard_device.restype = i32
    L.spz_amd_decode_shard_device.argtypes = [vp, sz, C.POINTER(Header), u64, u64, i32, C.POINTER(CloudPtrs), vp]
    L.spz_amd_decode_gather_device.restype = i32
    L.spz_amd_decode_gather_device.argtypes = [vp, sz, C.POINTER(Header), vp, u64, i32, C.POINTER(CloudPtrs), vp]
    L.spz_amd_median_scale_sum_device.restype = i32
    L.spz_amd_median_scale_sum_device.argtypes = [vp, u64, vp, vp, vp]
    L.spz_amd_median_scale_sum_host.restype = i32
    L.spz_amd_median_scale_sum_host.argtypes = [vp, u64, vp, i32]
    L.spz_amd_cloud_buffers_alloc.restype = i32
    L.spz_amd_cloud_buffers_alloc.argtypes = [u64, i32, i32, vp, i32, i32, vp, C.POINTER(CloudBuffers)]
    L.spz_amd_cloud_buffers_free.restype = i32
    L.spz_amd_cloud_buffers_free.argtypes = [C.POINTER(CloudBuffers)]
    L.spz_amd_decode_gather_host.restype = i32
    L.spz_amd_decode_gather_host.argtypes = [vp, sz, u64, vp, u64, i32, C.POINTER(CloudPtrs), i32]
    L.spz_amd_convert_coordinates_device.restype = i32
    L.spz_amd_convert_coordinates_device.argtypes = [vp, vp, vp, u64, i32, i32, i32, vp]
    L.spz_amd_encode_host.restype = i32
    L.spz_amd_encode_host.argtypes = [C.POINTER(CloudPtrs), u64, i32, i32, i32, i32, vp, sz, i32]
    L.spz_amd_decode_host.restype = i32
    L.spz_amd_decode_host.argtypes = [vp, sz, i32, C.POINTER(CloudPtrs), i32]
    L.spz_amd_decode_host_ex.restype = i32
    L.spz_amd_decode_host_ex.argtypes = [vp, sz, u64, i32, C.POINTER(CloudPtrs), i32]
    L.spz_amd_convert_coordinates_host.restype = i32
    L.spz_amd_convert_coordinates_host.argtypes = [vp, vp, vp, u64, i32, i32, i32, i32]
    L.spz_amd_get_tables.restype = i32
    L.spz_amd_get_tables.argtypes = [vp, vp, vp]
    L.spz_amd_ply_default_columns.restype = i32
    L.spz_amd_ply_default_columns.argtypes = [i32, C.POINTER(PlyColumns)]
    L.spz_amd_ply_rows_to_cloud_device.restype = i32
    L.spz_amd_ply_rows_to_cloud_device.argtypes = [vp, u64, C.POINTER(PlyColumns), i32, C.POINTER(CloudPtrs), vp]
    L.spz_amd_cloud_to_ply_rows_device.restype = i32
    L.spz_amd_cloud_to_ply_rows_device.argtypes = [C.POINTER(CloudPtrs), u64, i32, i32, vp, vp]
    L.spz_amd_ply_rows_to_cloud_host.restype = i32
    L.spz_amd_ply_rows_to_cloud_host.argtypes = [vp, u64, C.POINTER(PlyColumns), i32, C.POINTER(CloudPtrs), i32]
    L.spz_amd_cloud_to_ply_rows_host.restype = i32
    L.spz_amd_cloud_to_ply_rows_host.argtypes = [C.POINTER(CloudPtrs), u64, i32, i32, vp, i32]
    u32 = C.c_uint
    L.spz_amd_encode_shard_sections_device.restype = i32
    L.spz_amd_encode_shard_sections_device.argtypes = [C.POINTER(CloudPtrs), u64, u64, u64, i32, i32, i32, i32, i32, u32, vp, sz, vp]
    L.spz_amd_shard_fragments.restype = i32
    L.spz_amd_shard_fragments.argtypes = [u64, u64, u64, i32, i32, C.POINTER(Fragments)]
    L.spz_amd_rccl_available.restype = i32
    L.spz_amd_last_rccl_error.restype = i32
    L.spz_amd_rccl_unique_id.restype = i32
    L.spz_amd_rccl_unique_id.argtypes = [vp]
    L.spz_amd_rccl_comm_init.restype = i32
    L.spz_amd_rccl_comm_init.argtypes = [vp, i32, i32, C.POINTER(vp)]
    L.spz_amd_rccl_comm_destroy.restype = i32
    L.spz_amd_rccl_comm_destroy.argtypes = [vp]
    L.spz_amd_gatherv_rccl.restype = i32
    L.spz_amd_gatherv_rccl.argtypes = [vp, i32, i32, i32, C.POINTER(u64), C.POINTER(u64), i32, i32, vp, vp, u32, vp]
    L.spz_amd_scatterv_rccl.restype = i32
    L.spz_amd_scatterv_rccl.argtypes = [vp, i32, i32, i32, C.POINTER(u64), C.POINTER(u64), i32, i32, vp, vp, u32, vp]
    L.spz_amd_ipc_alloc.restype = i32
    L.spz_amd_ipc_alloc.argtypes = [sz, C.POINTER(vp), vp]
    L.spz_amd_ipc_free.restype = i32
    L.spz_amd_ipc_free.argtypes = [vp]
    L.spz_amd_ipc_open.restype = i32
    L.spz_amd_ipc_open.argtypes = [vp, C.POINTER(vp)]
    L.spz_amd_ipc_close.restype = i32
    L.spz_amd_ipc_close.argtypes = [vp]
    L.spz_amd_selftest_device.restype = i32
    L.spz_amd_selftest_device.argtypes = [i32, u64, u64, C.POINTER(u64 * 3), vp]
    L.spz_amd_zlib_parse_open.restype = i32
    L.spz_amd_zlib_parse_open.argtypes = [vp, u64, u64, vp, u32, i32, C.POINTER(vp), C.POINTER(u64), C.POINTER(u32)]
    L.spz_amd_zlib_parse_open_ex.restype = i32
    L.spz_amd_zlib_parse_open_ex.argtypes = [vp, u64, u64, vp, u32, i32, C.POINTER(vp), C.POINTER(u64), C.POINTER(u32), vp, vp]
    L.spz_amd_zlib_parse_fetch.restype = i32
    L.spz_amd_zlib_parse_fetch.argtypes = [vp, vp, vp]
    L.spz_amd_zlib_parse_close.restype = None
    L.spz_amd_zlib_parse_close.argtypes = [vp]
    L.spz_amd_zlib_parse_append.restype = i32
    L.spz_amd_zlib_parse_append.argtypes = [vp, vp, vp, u64]
    L.spz_amd_zlib_block_stats.restype = i32
    L.spz_amd_zlib_block_stats.argtypes = [vp, vp, u32, u32, vp, vp, vp, vp]
    L.spz_amd_decode_host_from_device.restype = i32
    L.spz_amd_decode_host_from_device.argtypes = [vp, sz, vp, i32, vp, i32]
    L.spz_amd_inflate_open.restype = i32
    L.spz_amd_inflate_open.argtypes = [vp, u64, i32, C.POINTER(vp), C.POINTER(u64)]
    L.spz_amd_inflate_crc_piece_bytes.restype = u32
    L.spz_amd_inflate_crc_piece_bytes.argtypes = []
    L.spz_amd_inflate_piece_crcs.restype = i32
    L.spz_amd_inflate_piece_crcs.argtypes = [vp, vp, u32, C.POINTER(u32)]
    L.spz_amd_inflate_fetch.restype = i32
    L.spz_amd_inflate_fetch.argtypes = [vp, vp]
    L.spz_amd_inflate_device_data.restype = vp
    L.spz_amd_inflate_device_data.argtypes = [vp]
    L.spz_amd_inflate_close.restype = None
    L.spz_amd_inflate_close.argtypes = [vp]
    L.spz_amd_zlib_encode_group.restype = i32
    L.spz_amd_zlib_encode_group.argtypes = [vp, vp, u32, u32, u32, u32, vp, vp, vp, u64, u64]
    L.spz_amd_zlib_encode_finish.restype = i32
    L.spz_amd_zlib_encode_finish.argtypes = [vp, u32, u64, vp, vp]
    L.spz_amd_zlib_block_trees.restype = i32
    L.spz_amd_zlib_block_trees.argtypes = [vp, u32, vp]
    L.spz_amd_zlib_encode_planned.restype = i32
    L.spz_amd_zlib_encode_planned.argtypes = [vp, vp, u32, u32, vp, u64]
    L.spz_amd_zlib_encode_finish_ex.restype = i32
    L.spz_amd_zlib_encode_finish_ex.argtypes = [vp, u32, u64, vp, vp, vp]
    L.spz_amd_zlib_encode_blocks.restype = i32
    L.spz_amd_zlib_encode_blocks.argtypes = [vp, vp, u32, u32, vp, vp, vp, u64, u64, vp, vp]
    return L


def status_string(status):
    return load_library().spz_amd_status_string(int(status)).decode()


def check(status, where):
    if status != OK:
        raise SpzAmdError(status, where)


def stream_layout(num_points, sh_degree, version=3):
    lay = Layout()
    check(load_library().spz_amd_stream_layout(int(num_points), int(sh_degree), int(version), C.byref(lay)),
          "spz_amd_stream_layout")
    return lay


def write_header(version, num_points, sh_degree, fractional_bits=12, antialiased=False):
    h = Header(int(version), int(num_points), int(sh_degree), int(fractional_bits), 1 if antialiased else 0, 0)
    out = (C.c_uint8 * 16)()
    check(load_library().spz_amd_write_header(C.byref(h), out), "spz_amd_write_header")
    return bytes(out)


def peek_header(stream_bytes, max_points=REFERENCE_MAX_POINTS):
    """stream_bytes: bytes-like HOST data (at least the header; the full stream for the size check).
    Returns (status, Header or None)."""
    buf = (C.c_uint8 * len(stream_bytes)).from_buffer_copy(stream_bytes) if len(stream_bytes) else None
    h = Header()
    rc = load_library().spz_amd_peek_header_ex(buf, len(stream_bytes), int(max_points), C.byref(h))
    return rc, (h if rc == OK else None)


def get_tables():
    import numpy as np
    a = np.zeros(256, np.float32)
    c = np.zeros(256, np.float32)
    t = np.zeros(255, np.float32)
    check(load_library().spz_amd_get_tables(a.ctypes.data, c.ctypes.data, t.ctypes.data), "spz_amd_get_tables")
    return a, c, t
