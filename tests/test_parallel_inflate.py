"""decompressGzipped's parallel reader for ORDINARY single-stream members (spz_amd/csrc/spz_inflate.cpp):
whatever it does, the result must be zlib's — same bytes for valid members from any writer and level,
same rejection for damaged ones (it verifies CRC-32 and ISIZE and otherwise hands over to the serial
readers).  Runs on the CPU; the parallel path is taken from 4 MiB and 8 threads."""
import gzip
import os
import zlib

import numpy as np
import pytest

import spz_amd.spz as spz
from test_exact_gzip import make


def zlib_verdict(b):
    d = zlib.decompressobj(31)
    try:
        out = d.decompress(b)
    except zlib.error:
        return None
    return out if d.eof else None


@pytest.fixture(autouse=True)
def eight_threads():
    old = os.environ.get("SPZ_AMD_GUNZIP_THREADS")
    os.environ["SPZ_AMD_GUNZIP_THREADS"] = "8"
    yield
    if old is None:
        del os.environ["SPZ_AMD_GUNZIP_THREADS"]
    else:
        os.environ["SPZ_AMD_GUNZIP_THREADS"] = old


def stream_like(rng, points):
    """Section mix of an SH3 stream: compressible position bytes, an incompressible block (stored blocks
    in the middle of the member), bucketed sh bytes."""
    pos = np.repeat(rng.integers(0, 256, 3 * points, dtype=np.uint8), 3)[: 9 * points]
    pos[::3] = rng.integers(0, 256, pos[::3].size, dtype=np.uint8)
    rot = rng.integers(0, 256, 4 * points, dtype=np.uint8)
    sh = np.clip(np.round(rng.normal(128, 20, 45 * points) / 8) * 8, 0, 255).astype(np.uint8)
    return pos.tobytes() + rot.tobytes() + sh.tobytes()


def parallel(gz):
    """Inflates through the public function and reports whether the parallel reader was the one that did it."""
    before = spz._parallel_inflate_count()
    out = spz._decompress_gzipped(gz)
    return out, spz._parallel_inflate_count() == before + 1


@pytest.mark.parametrize("level", [1, 6, 9])
def test_valid_members_from_zlib_at_several_levels(level):
    rng = np.random.default_rng(level)
    for data in (stream_like(rng, 150_000), make("nibbles", 11_000_000, rng)):
        co = zlib.compressobj(level, zlib.DEFLATED, 31, 9)
        gz = co.compress(data) + co.flush()
        assert len(gz) > (4 << 20)
        out, used = parallel(gz)
        assert out == data and used
    small = make("words", 3_000_000, rng)                  # below 4 MiB of member: the serial readers
    out, used = parallel(spz._compress_gzipped(small))
    assert out == small and not used


def test_other_writers_and_headers():
    rng = np.random.default_rng(11)
    data = stream_like(rng, 150_000)
    assert parallel(gzip.compress(data, 6)) == (data, True)                              # mtime, XFL, OS differ
    assert parallel(spz._compress_gzipped(data)) == (data, True)                          # the exact parallel writer
    one = spz._compress_gzipped(data)
    hdr = bytearray(one[:10]); hdr[3] = 0x08 | 0x10
    named = bytes(hdr) + b"scene.bin\0" + b"comment\0" + one[10:]
    assert parallel(named) == (data, True)
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)                         # static blocks only
    fixed = co.compress(data) + co.flush()
    assert parallel(fixed) == (data, False)       # no dynamic block to start a chunk at: serial
    co = zlib.compressobj(0, zlib.DEFLATED, 31)                                          # stored blocks only
    stored = co.compress(data) + co.flush()
    assert parallel(stored) == (data, False)
    # sync-flushed stream: empty stored blocks between Huffman blocks
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 9)
    parts = [co.compress(data[i:i + 700_000]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(data), 700_000)]
    flushed = b"".join(parts) + co.flush()
    assert parallel(flushed) == (data, True)


def test_damaged_members_get_zlibs_verdict():
    rng = np.random.default_rng(12)
    data = stream_like(rng, 130_000)
    gz = spz._compress_gzipped(data)
    assert len(gz) > (4 << 20)
    cases = {}
    for k, at in enumerate((len(gz) // 7, len(gz) - 1000)):
        b = bytearray(gz); b[at] ^= 0x04
        cases[f"flip{k}"] = bytes(b)
    b = bytearray(gz); b[-6] ^= 1
    cases["bad_crc"] = bytes(b)
    b = bytearray(gz); b[-2] ^= 1
    cases["bad_isize"] = bytes(b)
    cases["truncated"] = gz[:-1000]
    cases["cut_trailer"] = gz[:-3]
    cases["trailing_bytes"] = gz + b"\0" * 11
    cases["two_members"] = gz + gz
    for name, b in cases.items():
        want = zlib_verdict(b)
        got, used = parallel(b)
        assert not used, name                        # nothing damaged or irregular is ever accepted by the parallel reader
        assert got == want, f"{name}: {'accepted' if got is not None else 'rejected'}, zlib {'accepts' if want is not None else 'rejects'}"
    assert zlib_verdict(cases["trailing_bytes"]) == data and zlib_verdict(cases["flip0"]) is None
