"""Inflate of ordinary single-stream gzip members on the MI355X (spz_amd/csrc/spz_inflate_dev.hip behind
spz_amd_inflate_*): decompressGzipped must return exactly the bytes zlib returns, the device must have done the
work where it can (a counter says), and whatever it cannot do (stored-only data, damaged members) must come out
of the host readers with zlib's verdict."""
import os
import zlib

import numpy as np
import pytest

import spz_amd.spz as spz
from test_exact_gzip import make, zlib_gzip

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def force_device_inflate():
    old = os.environ.get("SPZ_AMD_GUNZIP_DEVICE")
    os.environ["SPZ_AMD_GUNZIP_DEVICE"] = "1"
    yield
    if old is None:
        os.environ.pop("SPZ_AMD_GUNZIP_DEVICE", None)
    else:
        os.environ["SPZ_AMD_GUNZIP_DEVICE"] = old


@pytest.mark.parametrize("kind", ["nibbles", "words", "runs", "sh_like"])
def test_device_inflate_returns_zlibs_bytes(kind):
    rng = np.random.default_rng(sum(kind.encode()) + 3)
    for n in (6_000_000, 9_000_001):
        data = make(kind, n, rng)
        for level in (6, 1, 9):
            co = zlib.compressobj(level, zlib.DEFLATED, 16 + 15, 9)
            member = co.compress(data) + co.flush()
            if len(member) < (1 << 20) + 64:
                continue    # below the forced threshold (highly compressible): the host readers' business
            before = spz._device_inflate_count()
            assert spz._decompress_gzipped(member) == data, f"{kind} n={n} level={level}"
            # every zlib level (profiles/r03_inflate_coverage.json: levels 1/6/9, memLevel 8 and 9, 120 members, none declined
            # for its block mix): a decline would still give the right bytes, but the reason must then be looked at
            assert spz._device_inflate_count() == before + 1, (f"{kind} n={n} level={level}: the device did not inflate it "
                                                               f"({spz._device_inflate_last_decline()!r})")


def test_device_inflate_of_a_real_stream_and_a_whole_load(tmp_path):
    from spz_amd.synth import make_cloud_numpy
    n, deg = 400_000, 3
    c = make_cloud_numpy(n, deg, 78)
    g = spz.GaussianCloud()
    g.sh_degree = deg
    for k in ("positions", "scales", "rotations", "alphas", "colors", "sh"):
        setattr(g, k, c[k])
    o = spz.PackOptions()
    o.from_coord = spz.RDF
    raw = spz._pack_to_stream(g, o)
    member = zlib_gzip(raw)
    before = spz._device_inflate_count()
    assert spz._decompress_gzipped(member) == raw
    assert spz._device_inflate_count() == before + 1
    path = str(tmp_path / "c.spz")
    open(path, "wb").write(member)
    u = spz.UnpackOptions()
    u.to_coord = spz.RDF
    back = spz.load_spz(path, u)
    assert back.num_points == n
    assert spz._device_inflate_count() == before + 2
    want = spz._unpack_from_stream(raw, u)
    for k in ("positions", "scales", "rotations", "alphas", "colors", "sh"):
        assert np.array_equal(np.asarray(getattr(back, k)).view(np.uint32), np.asarray(getattr(want, k)).view(np.uint32)), k


def test_what_the_device_declines_or_rejects_still_gets_zlibs_verdict():
    rng = np.random.default_rng(12)
    noise = make("bytes", 5_000_000, rng)               # stored blocks only: a parallel copy, the device's since round 3
    member = zlib_gzip(noise)
    before = spz._device_inflate_count()
    assert spz._decompress_gzipped(member) == noise
    assert spz._device_inflate_count() == before + 1, spz._device_inflate_last_decline()
    tiny = zlib_gzip(make("sh_like", 300_000, rng))     # far below the reader's size: not asked, and it says so
    assert spz._decompress_gzipped(tiny) is not None
    assert spz._device_inflate_count() == before + 1
    data = make("sh_like", 8_000_000, rng)
    member = bytearray(zlib_gzip(data))
    good = bytes(member)
    member[len(member) // 2] ^= 0x10                       # damage in the middle: zlib's verdict (an error)
    assert spz._decompress_gzipped(bytes(member)) is None
    member = bytearray(good)
    member[-6] ^= 0x01                                     # a wrong CRC-32 in the trailer
    assert spz._decompress_gzipped(bytes(member)) is None
    assert spz._decompress_gzipped(good) == data


def test_random_damage_gets_zlibs_verdict_not_a_crash():
    """Bit flips, truncations and spliced garbage anywhere in a member: whatever the device makes of it, the answer is
    zlib's (the bytes, or an error), because nothing is believed before the CRC-32 and ISIZE of the trailer match."""
    rng = np.random.default_rng(21)
    data = make("sh_like", 5_000_000, rng)
    good = zlib_gzip(data)
    for trial in range(12):
        m = bytearray(good)
        kind = trial % 3
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                m[int(rng.integers(10, len(m) - 8))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            m = m[: int(rng.integers(len(m) // 2, len(m) - 1))]
        else:
            at = int(rng.integers(10, len(m) - 70000))
            m[at:at + 65536] = rng.integers(0, 256, 65536, dtype=np.uint8).tobytes()
        try:
            want = zlib.decompress(bytes(m), 16 + 15)
        except zlib.error:
            want = None
        assert spz._decompress_gzipped(bytes(m)) == want, f"trial {trial}"


def test_stored_runs_between_huffman_sections():
    """Incompressible sections (zlib stores them: runs of stored blocks, whose raw bytes now and then read like a block
    header) between compressible ones — the shape of a real .spz stream.  The device walks the stored runs, copies
    them with a kernel of their own and drops look-alike block starts; it must still be the one that did the work."""
    rng = np.random.default_rng(31)
    parts = [make("sh_like", 3_000_000, rng), make("bytes", 3_000_000, rng), make("sh_like", 2_500_000, rng),
             make("bytes", 2_000_000, rng), make("words", 2_000_000, rng), make("bytes", 700_000, rng),
             make("nibbles", 1_500_000, rng)]
    data = b"".join(parts)
    member = zlib_gzip(data)
    before = spz._device_inflate_count()
    assert spz._decompress_gzipped(member) == data
    assert spz._device_inflate_count() == before + 1
