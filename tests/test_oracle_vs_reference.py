"""Pins the CPU restatement against the REFERENCE's own compiled C++ (oracle/_ref/libspz_ref.so)
on fresh random inputs.  Runs wherever that library exists (the build container, and the GPU box
since the built .so travels); skipped otherwise — the committed golden vectors then carry the pin."""
import os

import numpy as np
import pytest

from conftest import FIELDS, assert_bits_equal, assert_bytes_equal


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_random_clouds_all_coordinate_pairs(oracle, reference, deg):
    from spz_amd.synth import make_cloud_numpy
    n = 3001
    for frm in range(9):
        c = make_cloud_numpy(n, deg, 500 + 10 * deg + frm)
        so = oracle.pack(c, n, deg, True, frm)
        sr = reference.pack(c, n, deg, True, frm)
        assert_bytes_equal(so, sr, f"pack deg={deg} from={frm}")
        to = (frm * 5 + 3) % 9
        rc, uo = oracle.unpack(sr, to)
        ur = reference.unpack(sr, n, deg, to)
        assert rc == 0 and ur["num_points"] == n
        for k in FIELDS:
            assert_bits_equal(uo[k], ur[k], f"unpack deg={deg} to={to} {k}")


def test_quaternion_helpers(oracle, reference):
    import ctypes as C
    rng = np.random.default_rng(8)
    q = rng.standard_normal((20000, 4)).astype(np.float32)
    q[:2000] = np.round(q[:2000])
    q[np.all(q[:, :] == 0, axis=1)] = [0, 0, 0, 1]
    want = reference.pack_quat(q.reshape(-1), 6)
    conv = (C.c_float * 21)()
    oracle.lib.spzo_coordinate_converter.argtypes = [C.c_int, C.c_int, C.c_void_p]
    oracle.lib.spzo_coordinate_converter(6, 4, conv)
    oracle.lib.spzo_pack_quat_smallest_three.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    got = np.zeros(q.shape[0] * 4, np.uint8)
    for i in range(q.shape[0]):
        oracle.lib.spzo_pack_quat_smallest_three(got[4 * i:].ctypes.data, q[i].ctypes.data, conv)
    assert_bytes_equal(got, want, "packQuaternionSmallestThree")


def test_gzip_container_bytes(reference):
    """The host gzip wrapper of the product (spz::compressGzipped in libspz_host.so) emits the same
    bytes as the reference's (same zlib, same deflateInit2 parameters, load-spz.cc:186-214)."""
    import spz_amd.spz as spz
    rng = np.random.default_rng(3)
    for size in (0, 1, 16, 8191, 8192, 8193, 300_000, 3_000_000):
        data = (rng.integers(0, 256, size, dtype=np.uint16) >> (size % 3 + 2)).astype(np.uint8)
        want = reference.compress_gzipped(data).tobytes()
        assert want[:10] == bytes.fromhex("1f8b0800000000000003")
        got = spz._compress_gzipped(data.tobytes())
        assert got == want, f"gzip bytes differ for size {size}"
        assert spz._decompress_gzipped(got) == data.tobytes()
    assert spz._decompress_gzipped(b"This is not a valid SPZ file") is None
    assert spz._decompress_gzipped(want[:-5]) is None  # truncated stream


def test_half_to_float_all_values(oracle, reference):
    hs = np.arange(65536, dtype=np.uint32)
    a = np.array([oracle.lib.spzo_half_to_float(int(h)) for h in hs], np.float32)
    b = np.array([reference.half_to_float(int(h)) for h in hs], np.float32)
    assert_bits_equal(a, b, "halfToFloat")


def test_convert_coordinates(oracle, reference):
    from spz_amd.synth import make_cloud_numpy
    n, deg = 777, 3
    c = make_cloud_numpy(n, deg, 9)
    for frm, to in ((4, 6), (6, 7), (1, 8), (0, 3), (5, 5)):
        a = oracle.convert_coordinates(c["positions"], c["rotations"], c["sh"], n, deg, frm, to)
        b = reference.convert_coordinates(c["positions"], c["rotations"], c["sh"], n, deg, frm, to)
        for x, y, nm in zip(a, b, ("positions", "rotations", "sh")):
            assert_bits_equal(x, y, nm)


def test_parallel_gzip_is_read_by_the_reference(reference):
    """Opt-in multi-threaded gzip: not byte-identical to the reference's single
    deflate stream, but the reference's own loadSpz must read it to the same floats."""
    import gzip
    import spz_amd.spz as spz
    from spz_amd.synth import make_cloud_numpy
    n, deg = 40_000, 3   # 2.6 MB stream -> three 1 MiB deflate blocks
    c = make_cloud_numpy(n, deg, 99)
    stream = reference.pack(c, n, deg, True, 6).tobytes()
    par = spz._compress_gzipped_parallel(stream, 4)
    one = spz._compress_gzipped(stream)
    # same gzip member header as zlib's except FLG.FEXTRA: the piece index rides in subfield "SZ"
    assert par != one and par[:3] == one[:3] and par[3] == 4 and par[4:10] == one[4:10]
    assert par[12:14] == b"SZ" and int.from_bytes(par[10:12], "little") == 4 + int.from_bytes(par[14:16], "little")
    assert gzip.decompress(par) == stream and spz._decompress_gzipped(par) == stream
    a = reference.load_spz(np.frombuffer(par, np.uint8), n, deg, 7)
    b = reference.load_spz(np.frombuffer(one, np.uint8), n, deg, 7)
    assert a["num_points"] == n
    for k in FIELDS:
        assert_bits_equal(a[k], b[k], k)
    # threads <= 1 and small inputs fall back to the reference-identical single stream
    assert spz._compress_gzipped_parallel(stream, 1) == one
    assert spz._compress_gzipped_parallel(stream[:1000], 8) == spz._compress_gzipped(stream[:1000])


def test_gunzip_readers_agree_with_zlib(reference):
    """decompressGzipped has three readers (piece-parallel through the "SZ" index, libdeflate, the
    reference's zlib loop) and must return what zlib returns for every input: valid members from any
    writer, members with other header fields, damaged indexes (the index is advisory), corrupt,
    truncated and trailing bytes."""
    import gzip
    import subprocess
    import sys
    import zlib
    import spz_amd.spz as spz
    rng = np.random.default_rng(8)
    raw = (rng.integers(0, 256, 5_300_017, dtype=np.uint16) >> 3).astype(np.uint8).tobytes()

    def zlib_verdict(b):
        # the reference's loop (load-spz.cc:141-182): success only when inflate reached Z_STREAM_END
        d = zlib.decompressobj(31)
        try:
            out = d.decompress(b)
        except zlib.error:
            return None
        return out if d.eof else None

    par = spz._compress_gzipped_parallel(raw, 8)
    one = spz._compress_gzipped(raw)
    named = gzip.compress(raw, 6)                                  # python's writer: mtime set, XFL/OS differ
    co = zlib.compressobj(9, zlib.DEFLATED, 31)
    other = co.compress(raw) + co.flush()
    cases = {"indexed": par, "single": one, "python": named, "level9": other}
    # FNAME + FCOMMENT + FHCRC in front of the same deflate data
    body = one[10:]
    hdr = bytearray(one[:10]); hdr[3] = 0x08 | 0x10 | 0x02
    hdr += b"scene.bin\0" + b"a comment\0"
    hdr += (zlib.crc32(bytes(hdr)) & 0xffff).to_bytes(2, "little")
    cases["fname_fcomment_fhcrc"] = bytes(hdr) + body
    bad_hcrc = bytearray(cases["fname_fcomment_fhcrc"]); bad_hcrc[len(hdr) - 1] ^= 0x5a   # zlib rejects a wrong header CRC16
    hdr2 = bytearray(one[:10]); hdr2[3] = 0x02
    hdr2 += (zlib.crc32(bytes(hdr2)) & 0xffff).to_bytes(2, "little")
    cases["fhcrc_only"] = bytes(hdr2) + body
    # damaged index: a piece size off by one, an impossible block size, a wrong total
    for name, at, val in (("index_piece", 16 + 20, 1), ("index_block", 16 + 4, 0x40), ("index_total", 16 + 8, 3)):
        b = bytearray(par); b[at] ^= val
        cases[name] = bytes(b)
    for name, b in list(cases.items()):
        assert zlib_verdict(b) == raw, name
    cases["fhcrc_wrong"] = bytes(bad_hcrc)
    assert zlib_verdict(cases["fhcrc_wrong"]) is None
    b = bytearray(cases["fhcrc_only"]); b[10] ^= 1
    cases["fhcrc_only_wrong"] = bytes(b)
    mid = bytearray(par); mid[len(mid) // 2] ^= 0x10
    cases["corrupt_indexed"] = bytes(mid)
    mid = bytearray(one); mid[len(mid) // 2] ^= 0x10
    cases["corrupt_single"] = bytes(mid)
    crc = bytearray(one); crc[-6] ^= 1
    cases["bad_crc"] = bytes(crc)
    isz = bytearray(par); isz[-1] ^= 1
    cases["bad_isize_indexed"] = bytes(isz)
    cases["truncated_indexed"] = par[:-9]
    cases["truncated_single"] = one[: len(one) // 2]
    cases["trailing_indexed"] = par + b"\0" * 7
    cases["trailing_single"] = one + b"junk"
    cases["two_members"] = one + one          # zlib's inflate stops after the first member (load-spz.cc:141-182)
    cases["empty"] = b""
    cases["header_only"] = one[:10]
    cases["not_gzip"] = b"This is not a valid SPZ file"
    for name, b in cases.items():
        want = zlib_verdict(b)
        got = spz._decompress_gzipped(b)
        assert got == want, f"{name}: reader disagrees with zlib ({'accepts' if got is not None else 'rejects'})"
    # the same through the zlib-only configuration (no libdeflate, one thread), in a fresh process
    code = ("import sys, zlib, spz_amd.spz as spz\n"
            "b = open(sys.argv[1], 'rb').read()\n"
            "assert spz._decompress_gzipped(b) == zlib.decompressobj(31).decompress(b)\n")
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".gz") as f:
        f.write(par); f.flush()
        env = dict(os.environ, SPZ_AMD_NO_LIBDEFLATE="1", SPZ_AMD_GUNZIP_THREADS="1",
                   PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        subprocess.run([sys.executable, "-c", code, f.name], check=True, env=env, timeout=120)
