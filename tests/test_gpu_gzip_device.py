"""The exact gzip writer with its LZ77 parse on the MI355X (spz_amd/csrc/spz_lz77.hip behind
spz_amd_zlib_parse_*): compressGzipped must return, byte for byte, what zlib 1.2.11 returns with the
reference's parameters (load-spz.cc:186-214) — the same oracle as tests/test_exact_gzip.py — and the
device must actually have done the parse (a counter says which way it ran)."""
import os
import zlib

import numpy as np
import pytest

import spz_amd.spz as spz
from test_exact_gzip import make, zlib_gzip

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(zlib.ZLIB_RUNTIME_VERSION != "1.2.11", reason="the exact writer restates zlib 1.2.11")]


@pytest.fixture(autouse=True)
def force_device_parse():
    old = os.environ.get("SPZ_AMD_GZIP_DEVICE")
    os.environ["SPZ_AMD_GZIP_DEVICE"] = "1"
    yield
    os.environ.pop("SPZ_AMD_GZIP_DEVICE_HUFFMAN", None)
    os.environ.pop("SPZ_AMD_GZIP_DEVICE_TREES", None)
    if old is None:
        os.environ.pop("SPZ_AMD_GZIP_DEVICE", None)
    else:
        os.environ["SPZ_AMD_GZIP_DEVICE"] = old


@pytest.mark.parametrize("huffman", ["trees", "1", "0"])
@pytest.mark.parametrize("kind", ["nibbles", "bytes", "words", "runs", "sh_like"])
def test_device_parse_bytes_equal_zlib(kind, huffman):
    """Sizes around the job (8 KiB), tile (4 KiB), block (32767 symbols) and link-segment (512 KiB) boundaries
    of the kernels; with the Huffman stage on the device too (stored blocks: "bytes"; static and dynamic: the
    rest) — trees included ("trees", the default) or built by the host ("1") — and with only the parse there ("0")."""
    os.environ["SPZ_AMD_GZIP_DEVICE_HUFFMAN"] = "0" if huffman == "0" else "1"
    os.environ["SPZ_AMD_GZIP_DEVICE_TREES"] = "1" if huffman == "trees" else "0"
    rng = np.random.default_rng(sum(kind.encode()) + 1)
    for n in ((1 << 20), (1 << 20) + 1, 1_300_001, (1 << 21) + 32768, 3_000_017):
        data = make(kind, n, rng)
        before = spz._device_gzip_parse_count()
        got = spz._compress_gzipped(data)
        assert got == zlib_gzip(data), f"{kind} n={n}: differs from zlib"
        assert spz._device_gzip_parse_count() == before + 1, f"{kind} n={n}: the device did not do the parse"


def test_device_parse_of_a_real_stream_and_of_its_save(tmp_path):
    """A 400 k-point SH3 stream (26 MB: 400 jobs, 50 link segments) and the whole saveSpz path on top."""
    from spz_amd.synth import make_cloud_numpy
    n, deg = 400_000, 3
    c = make_cloud_numpy(n, deg, 77)
    g = spz.GaussianCloud()
    g.sh_degree = deg
    g.antialiased = True
    for k in ("positions", "scales", "rotations", "alphas", "colors", "sh"):
        setattr(g, k, c[k])
    o = spz.PackOptions()
    o.from_coord = spz.RDF
    raw = spz._pack_to_stream(g, o)
    before = spz._device_gzip_parse_count()
    member = spz._compress_gzipped(raw)
    assert spz._device_gzip_parse_count() == before + 1
    assert member == zlib_gzip(raw)
    path = str(tmp_path / "c.spz")
    assert spz.save_spz(g, o, path)
    assert open(path, "rb").read() == member


def test_device_parse_is_off_below_the_threshold_and_when_disabled():
    rng = np.random.default_rng(5)
    data = make("sh_like", 2_000_000, rng)
    want = zlib_gzip(data)
    os.environ.pop("SPZ_AMD_GZIP_DEVICE")           # default: 2 MiB and more
    before = spz._device_gzip_parse_count()
    assert spz._compress_gzipped(data) == want
    assert spz._device_gzip_parse_count() == before
    os.environ["SPZ_AMD_GZIP_DEVICE"] = "0"
    big = make("sh_like", 9_000_000, rng)
    assert spz._compress_gzipped(big) == zlib_gzip(big)
    assert spz._device_gzip_parse_count() == before
    os.environ["SPZ_AMD_GZIP_DEVICE"] = "1"


def test_device_declines_what_it_cannot_splice_and_the_result_is_still_zlibs():
    """Constant input: neighbouring jobs never reach the same lazy-match state (258-byte matches out of phase), at
    any job size; the device path declines, the host writer declines too, zlib itself writes the member."""
    data = bytes(3 << 20)
    before = spz._device_gzip_parse_count()
    assert spz._compress_gzipped(data) == zlib_gzip(data)
    assert spz._device_gzip_parse_count() == before


def test_long_repeats_need_larger_jobs_and_get_them():
    """Pieces of up to 40 KiB that come back at distances around the 32 KiB window: with 8 and 16 KiB jobs two neighbours
    often do not meet inside the successor's range; the stage is rerun with larger jobs on the device."""
    rng = np.random.default_rng(9)
    pieces = [rng.integers(0, 256, int(rng.integers(1 << 10, 40 << 10)), dtype=np.uint8).tobytes() for _ in range(6)]
    out = bytearray()
    while len(out) < (3 << 20):
        out += pieces[int(rng.integers(0, 6))]
        out += rng.integers(0, 64, int(rng.integers(0, 3000)), dtype=np.uint8).tobytes()
    data = bytes(out)
    before = spz._device_gzip_parse_count()
    assert spz._compress_gzipped(data) == zlib_gzip(data)
    assert spz._device_gzip_parse_count() == before + 1


@pytest.mark.parametrize("deg", [0, 1, 2])
def test_whole_files_of_every_degree_through_the_device_container(deg, tmp_path):
    """saveSpz / loadSpz with both halves of the container stage forced onto the device, SH degrees 0-2 (degree 3 is
    the test above and the bench): the file is zlib's member of the raw stream, and it loads back to the same floats
    as the stream decodes to."""
    from spz_amd.synth import make_cloud_numpy
    n = 300_000
    c = make_cloud_numpy(n, deg, 90 + deg)
    g = spz.GaussianCloud()
    g.sh_degree = deg
    for k in ("positions", "scales", "rotations", "alphas", "colors", "sh"):
        if len(c[k]):
            setattr(g, k, c[k])
    o = spz.PackOptions()
    o.from_coord = spz.RUF
    raw = spz._pack_to_stream(g, o)
    path = str(tmp_path / f"d{deg}.spz")
    old = os.environ.get("SPZ_AMD_GUNZIP_DEVICE")
    os.environ["SPZ_AMD_GUNZIP_DEVICE"] = "1"
    try:
        parses = spz._device_gzip_parse_count()
        assert spz.save_spz(g, o, path)
        assert spz._device_gzip_parse_count() == parses + 1
        assert open(path, "rb").read() == zlib_gzip(raw)
        u = spz.UnpackOptions()
        u.to_coord = spz.RUF
        back = spz.load_spz(path, u)
        want = spz._unpack_from_stream(raw, u)
        assert back.num_points == n and back.sh_degree == deg
        for k in ("positions", "scales", "rotations", "alphas", "colors", "sh"):
            assert np.array_equal(np.asarray(getattr(back, k)).view(np.uint32), np.asarray(getattr(want, k)).view(np.uint32)), k
    finally:
        if old is None:
            os.environ.pop("SPZ_AMD_GUNZIP_DEVICE", None)
        else:
            os.environ["SPZ_AMD_GUNZIP_DEVICE"] = old
