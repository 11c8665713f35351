"""ctypes front-ends for the two CPU checkers.  TEST INFRASTRUCTURE ONLY.

* ``Oracle``    -> oracle/liboracle.so        (plain-C restatement, spz_oracle.c)
* ``Reference`` -> oracle/_ref/libspz_ref.so  (the reference's own C++ + ref_shim.cc)

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may
import this module; the product package ``spz_amd`` never does.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libspz_ref.so")

_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)


def sh_dim(deg):
    return {0: 0, 1: 3, 2: 8, 3: 15}[int(deg)]


def stream_size(n, deg, version=3):
    pos_b = 6 if version == 1 else 9
    rot_b = 4 if version >= 3 else 3
    return 16 + n * (pos_b + 1 + 3 + 3 + rot_b + sh_dim(deg) * 3)


def _fp(a):
    return a.ctypes.data_as(_f32p) if a is not None and a.size else C.cast(None, _f32p)


def _bp(a):
    return a.ctypes.data_as(_u8p) if a is not None and a.size else C.cast(None, _u8p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1)


def _alloc_cloud(n, deg):
    return dict(
        positions=np.zeros(n * 3, np.float32),
        scales=np.zeros(n * 3, np.float32),
        rotations=np.zeros(n * 4, np.float32),
        alphas=np.zeros(n, np.float32),
        colors=np.zeros(n * 3, np.float32),
        sh=np.zeros(n * sh_dim(deg) * 3, np.float32),
    )


def _cloud_ptrs(cloud):
    return [_fp(_f32(cloud[k])) for k in ("positions", "scales", "rotations", "alphas", "colors", "sh")]


class Oracle:
    """Plain-C restatement (liboracle.so)."""

    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = self.lib = C.CDLL(path)
        L.spzo_stream_size.restype = C.c_size_t
        L.spzo_stream_size.argtypes = [C.c_int64, C.c_int, C.c_int]
        L.spzo_pack.restype = C.c_size_t
        L.spzo_pack.argtypes = [_f32p] * 6 + [C.c_int32, C.c_int, C.c_int, C.c_int, C.c_int, _u8p]
        L.spzo_peek.restype = C.c_int
        L.spzo_peek.argtypes = [_u8p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)] + [C.POINTER(C.c_int)] * 3
        L.spzo_unpack.restype = C.c_int
        L.spzo_unpack.argtypes = [_u8p, C.c_size_t, C.c_int] + [_f32p] * 6
        L.spzo_convert_coordinates.restype = None
        L.spzo_convert_coordinates.argtypes = [_f32p] * 3 + [C.c_int32, C.c_int, C.c_int, C.c_int]
        L.spzo_alpha_byte.restype = C.c_uint8
        L.spzo_alpha_byte.argtypes = [C.c_float]
        L.spzo_alpha_value.restype = C.c_float
        L.spzo_alpha_value.argtypes = [C.c_uint8]
        L.spzo_color_value.restype = C.c_float
        L.spzo_color_value.argtypes = [C.c_uint8]
        L.spzo_half_to_float.restype = C.c_float
        L.spzo_half_to_float.argtypes = [C.c_uint16]

    def pack(self, cloud, n, deg, antialiased=False, from_coord=0, version=3):
        arrs = {k: _f32(cloud[k]) for k in ("positions", "scales", "rotations", "alphas", "colors", "sh")}
        out = np.zeros(stream_size(n, deg, version), np.uint8)
        w = self.lib.spzo_pack(*[_fp(arrs[k]) for k in ("positions", "scales", "rotations", "alphas", "colors", "sh")],
                               n, deg, int(antialiased), from_coord, version, _bp(out))
        if w != out.size:
            raise RuntimeError(f"spzo_pack wrote {w}, expected {out.size}")
        return out

    def peek(self, stream):
        stream = np.ascontiguousarray(stream, np.uint8)
        v, n = C.c_uint32(), C.c_uint32()
        d, fb, aa = C.c_int(), C.c_int(), C.c_int()
        rc = self.lib.spzo_peek(_bp(stream), stream.size, v, n, d, fb, aa)
        if rc:
            return rc, None
        return 0, dict(version=v.value, num_points=n.value, sh_degree=d.value, fractional_bits=fb.value,
                       antialiased=bool(aa.value))

    def unpack(self, stream, to_coord=0):
        stream = np.ascontiguousarray(stream, np.uint8)
        rc, h = self.peek(stream)
        if rc:
            return rc, None
        out = _alloc_cloud(h["num_points"], h["sh_degree"])
        rc = self.lib.spzo_unpack(_bp(stream), stream.size, to_coord, *_cloud_ptrs(out))
        out.update(num_points=h["num_points"], sh_degree=h["sh_degree"], antialiased=h["antialiased"])
        return rc, out

    def convert_coordinates(self, positions, rotations, sh, n, deg, from_coord, to_coord):
        p, r, s = _f32(positions).copy(), _f32(rotations).copy(), _f32(sh).copy()
        self.lib.spzo_convert_coordinates(_fp(p), _fp(r), _fp(s), n, sh_dim(deg), from_coord, to_coord)
        return p, r, s

    def ply_rows_to_cloud(self, rows, n, cols, to_coord=0):
        """cols: a ctypes structure with spz_amd_ply_columns' layout (e.g. spz_amd.abi.PlyColumns)."""
        rows = _f32(rows)
        d = cols.sh_dim * 3
        out = dict(positions=np.zeros(n * 3, np.float32), scales=np.zeros(n * 3, np.float32),
                   rotations=np.zeros(n * 4, np.float32), alphas=np.zeros(n, np.float32),
                   colors=np.zeros(n * 3, np.float32), sh=np.zeros(n * d, np.float32))
        self.lib.spzo_ply_rows_to_cloud.restype = None
        self.lib.spzo_ply_rows_to_cloud.argtypes = [_f32p, C.c_int32, C.c_void_p, C.c_int] + [_f32p] * 6
        self.lib.spzo_ply_rows_to_cloud(_fp(rows), n, C.byref(cols), to_coord, *_cloud_ptrs(out))
        return out

    def cloud_to_ply_rows(self, cloud, n, sh_dim, from_coord=0):
        arrs = [_f32(cloud[k]) for k in ("positions", "scales", "rotations", "alphas", "colors", "sh")]
        rows = np.zeros(n * (17 + 3 * sh_dim), np.float32)
        self.lib.spzo_cloud_to_ply_rows.restype = None
        self.lib.spzo_cloud_to_ply_rows.argtypes = [_f32p] * 6 + [C.c_int32, C.c_int, C.c_int, _f32p]
        self.lib.spzo_cloud_to_ply_rows(*[_fp(a) for a in arrs], n, sh_dim, from_coord, _fp(rows))
        return rows

    def median_volume(self, scales, n):
        a = _f32(scales)
        self.lib.spzo_median_volume.restype = C.c_float
        self.lib.spzo_median_volume.argtypes = [_f32p, C.c_int32]
        return np.float32(self.lib.spzo_median_volume(_fp(a), n))

    def alpha_decode_table(self):
        return np.array([self.lib.spzo_alpha_value(b) for b in range(256)], np.float32)

    def color_decode_table(self):
        return np.array([self.lib.spzo_color_value(b) for b in range(256)], np.float32)

    def alpha_byte(self, a):
        return self.lib.spzo_alpha_byte(float(np.float32(a)))


class Reference:
    """The reference's own C++ (oracle/_ref/libspz_ref.so)."""

    def __init__(self, path=REF_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing (built by `make -C oracle` where /root/reference exists)")
        L = self.lib = C.CDLL(path)
        cloud6 = [_f32p] * 6
        L.ref_pack.restype = C.c_size_t
        L.ref_pack.argtypes = cloud6 + [C.c_int32, C.c_int, C.c_int, C.c_int, _u8p, C.c_size_t]
        L.ref_unpack.restype = C.c_int
        L.ref_unpack.argtypes = [_u8p, C.c_size_t, C.c_int] + cloud6 + [C.POINTER(C.c_int32)]
        L.ref_save_spz.restype = C.c_size_t
        L.ref_save_spz.argtypes = cloud6 + [C.c_int32, C.c_int, C.c_int, C.c_int, _u8p, C.c_size_t]
        L.ref_load_spz.restype = C.c_int
        L.ref_load_spz.argtypes = [_u8p, C.c_int32, C.c_int] + cloud6 + [C.POINTER(C.c_int32)]
        L.ref_compress_gzipped.restype = C.c_size_t
        L.ref_compress_gzipped.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t]
        L.ref_converter.restype = None
        L.ref_converter.argtypes = [C.c_int, C.c_int, _f32p, _f32p, _f32p]
        L.ref_convert_coordinates.restype = None
        L.ref_convert_coordinates.argtypes = [_f32p] * 3 + [C.c_int32, C.c_int, C.c_int, C.c_int]
        L.ref_pack_quat.restype = None
        L.ref_pack_quat.argtypes = [_f32p, C.c_int32, C.c_int, _u8p]
        L.ref_unpack_quat_smallest_three.restype = None
        L.ref_unpack_quat_smallest_three.argtypes = [_u8p, C.c_int32, C.c_int, _f32p]
        L.ref_unpack_quat_first_three.restype = None
        L.ref_unpack_quat_first_three.argtypes = [_u8p, C.c_int32, C.c_int, _f32p]
        L.ref_half_to_float.restype = C.c_float
        L.ref_half_to_float.argtypes = [C.c_uint16]
        L.ref_bench_pack_unpack.restype = C.c_int
        L.ref_bench_pack_unpack.argtypes = cloud6 + [C.c_int32, C.c_int, C.c_int, C.c_int,
                                                     C.POINTER(C.c_double), C.POINTER(C.c_double), _u8p, C.c_size_t,
                                                     C.POINTER(C.c_uint64)]

    def pack(self, cloud, n, deg, antialiased=False, from_coord=0):
        arrs = [_f32(cloud[k]) for k in ("positions", "scales", "rotations", "alphas", "colors", "sh")]
        out = np.zeros(stream_size(n, deg, 3), np.uint8)
        w = self.lib.ref_pack(*[_fp(a) for a in arrs], n, deg, int(antialiased), from_coord, _bp(out), out.size)
        if w != out.size:
            raise RuntimeError(f"ref_pack wrote {w}, expected {out.size}")
        return out

    def unpack(self, stream, n, deg, to_coord=0):
        """n/deg: expected sizes for allocation (caller knows them from the header)."""
        stream = np.ascontiguousarray(stream, np.uint8)
        out = _alloc_cloud(n, deg)
        info = (C.c_int32 * 3)()
        self.lib.ref_unpack(_bp(stream), stream.size, to_coord, *_cloud_ptrs(out), info)
        out.update(num_points=info[0], sh_degree=info[1], antialiased=bool(info[2]))
        return out

    def save_spz(self, cloud, n, deg, antialiased=False, from_coord=0):
        arrs = [_f32(cloud[k]) for k in ("positions", "scales", "rotations", "alphas", "colors", "sh")]
        cap = stream_size(n, deg, 3) + (stream_size(n, deg, 3) >> 8) + 1024
        out = np.zeros(cap, np.uint8)
        w = self.lib.ref_save_spz(*[_fp(a) for a in arrs], n, deg, int(antialiased), from_coord, _bp(out), cap)
        if w == 0:
            raise RuntimeError("ref_save_spz failed")
        return out[:w].copy()

    def load_spz(self, gz, n, deg, to_coord=0):
        gz = np.ascontiguousarray(gz, np.uint8)
        out = _alloc_cloud(n, deg)
        info = (C.c_int32 * 3)()
        self.lib.ref_load_spz(_bp(gz), gz.size, to_coord, *_cloud_ptrs(out), info)
        out.update(num_points=info[0], sh_degree=info[1], antialiased=bool(info[2]))
        return out

    def compress_gzipped(self, data):
        data = np.ascontiguousarray(data, np.uint8)
        cap = data.size + (data.size >> 8) + 1024
        out = np.zeros(cap, np.uint8)
        w = self.lib.ref_compress_gzipped(_bp(data), data.size, _bp(out), cap)
        if w == 0:
            raise RuntimeError("ref_compress_gzipped failed")
        return out[:w].copy()

    def converter(self, from_coord, to_coord):
        p, q, s = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(15, np.float32)
        self.lib.ref_converter(from_coord, to_coord, _fp(p), _fp(q), _fp(s))
        return p, q, s

    def convert_coordinates(self, positions, rotations, sh, n, deg, from_coord, to_coord):
        p, r, s = _f32(positions).copy(), _f32(rotations).copy(), _f32(sh).copy()
        self.lib.ref_convert_coordinates(_fp(p), _fp(r), _fp(s), n, deg, from_coord, to_coord)
        return p, r, s

    def pack_quat(self, q, from_coord=0):
        q = _f32(q)
        n = q.size // 4
        out = np.zeros(n * 4, np.uint8)
        self.lib.ref_pack_quat(_fp(q), n, from_coord, _bp(out))
        return out

    def unpack_quat_smallest_three(self, r, to_coord=0):
        r = np.ascontiguousarray(r, np.uint8).reshape(-1)
        n = r.size // 4
        out = np.zeros(n * 4, np.float32)
        self.lib.ref_unpack_quat_smallest_three(_bp(r), n, to_coord, _fp(out))
        return out

    def unpack_quat_first_three(self, r, to_coord=0):
        r = np.ascontiguousarray(r, np.uint8).reshape(-1)
        n = r.size // 3
        out = np.zeros(n * 4, np.float32)
        self.lib.ref_unpack_quat_first_three(_bp(r), n, to_coord, _fp(out))
        return out

    def half_to_float(self, h):
        return self.lib.ref_half_to_float(int(h))

    def save_ply(self, cloud, n, deg, from_coord, filename):
        arrs = [_f32(cloud[k]) for k in ("positions", "scales", "rotations", "alphas", "colors", "sh")]
        self.lib.ref_save_ply.restype = C.c_int
        self.lib.ref_save_ply.argtypes = [_f32p] * 6 + [C.c_int32, C.c_int, C.c_int, C.c_char_p]
        return self.lib.ref_save_ply(*[_fp(a) for a in arrs], n, deg, from_coord, filename.encode())

    def load_ply(self, filename, n, deg, to_coord=0):
        out = _alloc_cloud(n, deg)
        info = (C.c_int32 * 3)()
        self.lib.ref_load_ply.restype = C.c_int
        self.lib.ref_load_ply.argtypes = [C.c_char_p, C.c_int] + [_f32p] * 6 + [C.POINTER(C.c_int32)]
        self.lib.ref_load_ply(filename.encode(), to_coord, *_cloud_ptrs(out), info)
        out.update(num_points=info[0], sh_degree=info[1], sh_size=info[2])
        return out

    def median_volume(self, scales, n):
        a = _f32(scales)
        self.lib.ref_median_volume.restype = C.c_float
        self.lib.ref_median_volume.argtypes = [_f32p, C.c_int32]
        return np.float32(self.lib.ref_median_volume(_fp(a), n))

    def packed_unpack(self, stream, indices, from_coord, to_coord):
        """PackedGaussians::at(i) bytes [count,65] and ::unpack(i, converter) floats [count,59] of a raw stream."""
        stream = np.ascontiguousarray(stream, np.uint8)
        idx = np.ascontiguousarray(indices, np.int32)
        b = np.zeros((idx.size, 65), np.uint8)
        f = np.zeros((idx.size, 59), np.float32)
        self.lib.ref_packed_unpack.restype = C.c_int
        self.lib.ref_packed_unpack.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32, C.c_int, C.c_int,
                                               C.c_void_p, C.c_void_p]
        rc = self.lib.ref_packed_unpack(stream.ctypes.data, stream.size, idx.ctypes.data, idx.size, from_coord,
                                        to_coord, b.ctypes.data, f.ctypes.data)
        if rc:
            raise RuntimeError(f"ref_packed_unpack rc={rc}")
        return b, f

    def bench_pack_unpack(self, cloud, n, deg, from_coord=0, to_coord=0, want_stream=False):
        arrs = [_f32(cloud[k]) for k in ("positions", "scales", "rotations", "alphas", "colors", "sh")]
        tp, tu = C.c_double(), C.c_double()
        stream = np.zeros(stream_size(n, deg, 3), np.uint8) if want_stream else None
        sums = (C.c_uint64 * 6)()
        rc = self.lib.ref_bench_pack_unpack(*[_fp(a) for a in arrs], n, deg, from_coord, to_coord,
                                            C.byref(tp), C.byref(tu), _bp(stream),
                                            stream.size if want_stream else 0, sums)
        if rc:
            raise RuntimeError(f"ref_bench_pack_unpack rc={rc}")
        self.last_decoded_bit_sums = [int(x) for x in sums]
        return tp.value, tu.value, stream
