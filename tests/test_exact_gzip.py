"""The multi-threaded gzip writer of the C++ layer (spz_amd/csrc/spz_deflate.cpp) must produce, byte for
byte, what zlib 1.2.11 produces with the reference's parameters (compressGzipped, load-spz.cc:186-214:
default level, gzip wrapper, memLevel 9, one Z_FINISH call) — it is what keeps `.spz` files identical to
the reference's while the container stage runs on all cores.  zlib itself (through Python's binding of
the same system library) is the oracle here; the reference build is checked on one large input too."""
import os
import zlib

import numpy as np
import pytest

import spz_amd.spz as spz

pytestmark = pytest.mark.skipif(zlib.ZLIB_RUNTIME_VERSION != "1.2.11",
                                reason="the exact writer restates zlib 1.2.11 and stands down for any other version")


def zlib_gzip(b):
    co = zlib.compressobj(-1, zlib.DEFLATED, 16 + 15, 9, zlib.Z_DEFAULT_STRATEGY)
    return co.compress(b) + co.flush()


def make(kind, n, rng):
    if kind == "nibbles":
        return rng.integers(0, 16, n, dtype=np.uint8).tobytes()
    if kind == "bytes":
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == "words":
        words = [bytes(rng.integers(97, 123, rng.integers(2, 9), dtype=np.uint8)) for _ in range(400)]
        out = bytearray()
        while len(out) < n:
            out += words[int(rng.integers(0, 400))] + b" "
        return bytes(out[:n])
    if kind == "runs":
        out = bytearray()
        while len(out) < n:
            out += bytes([int(rng.integers(0, 4))]) * int(rng.integers(1, 2000))
        return bytes(out[:n])
    if kind == "sh_like":       # bucketed values around 128, like the sh section of a stream
        return (np.clip(np.round(rng.normal(128, 20, n) / 8) * 8, 0, 255)).astype(np.uint8).tobytes()
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["nibbles", "bytes", "words", "runs", "sh_like"])
def test_bytes_equal_zlib(kind):
    """Several parse jobs (4-window chunks: 128 KiB, so every input below is split) and the single-job
    form, with the per-loop-top window-phase check switched on."""
    os.environ["SPZ_AMD_EXACT_GZIP_CHECK"] = "1"
    rng = np.random.default_rng(sum(kind.encode()))
    for n in (131072, 131073, 200001, 262144, (1 << 20) + 12345):
        data = make(kind, n, rng)
        want = zlib_gzip(data)
        for windows, threads in ((4, 4), (8, 3), (4096, 1)):
            got = spz._compress_gzipped_exact(data, threads, windows)
            assert got == want, f"{kind} n={n} windows={windows}: differs from zlib"


def test_block_count_edges():
    """Sizes for which the symbol count passes through multiples of 32767 (a block is flushed when the
    symbol buffer fills): the empty final block, and the full block that becomes the final one when the
    literal pending at the end of the input is what fills the buffer."""
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, 200000, dtype=np.uint8).tobytes()
    empty_final = 0
    for n in list(range(163835, 163990)) + list(range(131072, 131080)):
        want = zlib_gzip(base[:n])
        assert spz._compress_gzipped_exact(base[:n], 2, 4) == want, n
        empty_final += want[-10:-8] == b"\x03\x00"
    assert empty_final >= 1        # the sweep did cross an exact multiple


def test_constant_input_is_declined_and_the_public_function_falls_back():
    """Neighbouring parse jobs must meet in the same matcher state inside their 32 KiB overlap; a run of
    one byte value longer than that never lets them (258-byte matches in different phases), so the writer
    declines and compressGzipped uses zlib: same bytes either way."""
    data = bytes(9 << 20)
    assert spz._compress_gzipped_exact(data, 4, 4) is None
    assert spz._compress_gzipped_exact(data, 1, 4096) == zlib_gzip(data)      # one job: nothing to meet
    assert spz._compress_gzipped(data) == zlib_gzip(data)


def test_inputs_outside_the_writer_are_declined():
    assert spz._compress_gzipped_exact(b"x" * 1000, 4, 4) is None              # below 128 KiB
    assert spz._compress_gzipped_exact(bytes(200000), 0, 4) is None            # no threads
    assert spz._compress_gzipped_exact(bytes(200000), 4, 2) is None            # chunks of fewer than 4 windows


def test_self_check_against_zlib_prefix():
    rng = np.random.default_rng(9)
    data = make("sh_like", 3 << 20, rng)
    want = zlib_gzip(data)
    assert spz._compress_gzipped_exact(data, 4, 8, 256 << 10) == want           # prefix check passes
    assert spz._compress_gzipped_exact(data, 4, 8, 1 << 30) == want             # prefix = whole input


def test_default_save_path_on_a_real_stream_equals_zlib_and_the_reference(reference):
    """A 19.5 MB SH3 stream and a 1.3 MB one (above the 1 MiB switch-over of compressGzipped): the default container bytes
    equal zlib's and the reference's own compressGzipped."""
    from spz_amd.synth import make_cloud_numpy
    n = 300_000
    c = make_cloud_numpy(n, 3, 21)
    stream = reference.pack(c, n, 3, False, 6)
    raw = stream.tobytes()
    got = spz._compress_gzipped(raw)
    assert got == zlib_gzip(raw)
    assert got == reference.compress_gzipped(stream).tobytes()
    assert spz._decompress_gzipped(got) == raw
    small = raw[: 16 + 65 * 20000]
    assert spz._compress_gzipped(small) == zlib_gzip(small)
    os.environ["SPZ_AMD_GZIP_EXACT_THREADS"] = "1"                               # zlib only
    try:
        assert spz._compress_gzipped(raw) == got
    finally:
        del os.environ["SPZ_AMD_GZIP_EXACT_THREADS"]


@pytest.mark.parametrize("kind", ["nibbles", "bytes", "words", "runs", "sh_like"])
def test_staged_parse_model_equals_zlib(kind):
    """The parse the MI355X runs (spz_lz77.hip), as its serial host model with the same stage functions and job
    geometry (spz_lz77_core.hpp / spz_lz77_model.cpp): hash2 links + zlib-chain ranks, the two match tables per
    position, the lazy state machine over 8 KiB jobs, the stitch into the successor's recorded states and the
    splice with the host's serial tail job.  Sizes around the 8 KiB job / 32 KiB window boundaries."""
    rng = np.random.default_rng(sum(kind.encode()) + 2)
    for n in (524288, 524289, 540673, 700001, (1 << 20) + 12345):
        data = make(kind, n, rng)
        got = spz._compress_gzipped_exact_model(data, 4, 256 << 10)
        assert got == zlib_gzip(data), f"{kind} n={n}: the staged parse differs from zlib"


def test_staged_parse_model_declines_what_it_cannot_splice():
    """Constant input: two neighbouring jobs never reach the same lazy-match state (their 258-byte matches stay out
    of phase), so the staged parse declines and the caller uses another path."""
    assert spz._compress_gzipped_exact_model(bytes(1 << 20), 4) is None
    assert spz._compress_gzipped_exact_model(b"x" * 1000, 4) is None           # below 512 KiB
