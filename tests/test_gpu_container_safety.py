"""The container stage on the device is the default for every saveSpz of 2 MiB and more (and loadSpz of large files) (spz_lz77.hip,
spz_inflate_dev.hip).  These tests are about what protects a user there: a wrong symbol out of the parse kernels must
never reach a file (the on-device symbol check, always on), SPZ_AMD_GZIP_VERIFY=1 must inflate the member on the
device and compare it with the input, the default route must give zlib's bytes at the BASELINE size (10 M SH3 points,
650 MB of stream), and several callers at once must each get their own right answer.
Reference behaviour to match: compressGzipped's bytes (load-spz.cc:186-214) and the reader's verdict (:141-184)."""
import os
import threading
import zlib

import numpy as np
import pytest

import spz_amd.spz as spz
from conftest import FIELDS
from test_exact_gzip import make, zlib_gzip

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(zlib.ZLIB_RUNTIME_VERSION != "1.2.11", reason="the exact writer restates zlib 1.2.11")]


class Env:
    """Environment variables for the duration of a with-block (the library reads them at every call)."""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = os.environ.get(k)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def cloud_of(n, deg, seed, antialiased=False):
    from spz_amd.synth import make_cloud_numpy
    c = make_cloud_numpy(n, deg, seed)
    g = spz.GaussianCloud()
    g.sh_degree = deg
    g.antialiased = antialiased
    for k in FIELDS:
        if len(c[k]):
            setattr(g, k, c[k])
    return g


def same_cloud(a, b):
    if a.num_points != b.num_points or a.sh_degree != b.sh_degree or a.antialiased != b.antialiased:
        return False
    return all(np.array_equal(np.asarray(getattr(a, k)).view(np.uint32), np.asarray(getattr(b, k)).view(np.uint32)) for k in FIELDS)


@pytest.mark.parametrize("trees", ["1", "0"])
def test_a_wrong_symbol_never_reaches_the_member(trees):
    """SPZ_AMD_TEST_CORRUPT_SYMBOL changes one symbol on the device BEFORE the block statistics, so trees, layout and
    bit counts are all consistent with it and (past the 256 KiB prefix that is compared with zlib) nothing the writer
    checked in round 2 could notice.  The symbol check must: the member is discarded, counted, logged, and the caller
    still gets zlib's bytes (from the host writer)."""
    rng = np.random.default_rng(11)
    data = make("sh_like", 3_000_017, rng)
    want = zlib_gzip(data)
    with Env(SPZ_AMD_GZIP_DEVICE="1", SPZ_AMD_GZIP_DEVICE_TREES=trees, SPZ_AMD_GZIP_VERIFY=None):
        parses = spz._device_gzip_parse_count()
        assert spz._compress_gzipped(data) == want          # the route works when nothing is wrong
        assert spz._device_gzip_parse_count() == parses + 1
        for index in (0, 7, 40_000, 1_000_003, 2_400_000):  # first block, inside the prefix, far behind it
            with Env(SPZ_AMD_TEST_CORRUPT_SYMBOL=index):
                rejects, parses = spz._device_gzip_reject_count(), spz._device_gzip_parse_count()
                got = spz._compress_gzipped(data)
                assert got == want, f"symbol {index}: the caller did not get zlib's member"
                assert spz._device_gzip_reject_count() == rejects + 1, f"symbol {index}: not caught by the symbol check"
                assert spz._device_gzip_parse_count() == parses, f"symbol {index}: the device member was returned"


def test_verify_level_1_inflates_on_the_device_and_level_2_compares_with_zlib():
    """SPZ_AMD_GZIP_VERIFY=1: the body is inflated where it lies (device reader) and compared with the input; a member
    the device reader declines (too small for it) is inflated on the host instead — the member is returned either way."""
    rng = np.random.default_rng(12)
    for kind, n in (("sh_like", 12_000_000), ("words", 9_000_000), ("nibbles", 1_500_000)):
        data = make(kind, n, rng)
        want = zlib_gzip(data)
        for level in ("1", "2"):
            with Env(SPZ_AMD_GZIP_DEVICE="1", SPZ_AMD_GZIP_VERIFY=level):
                parses, rejects = spz._device_gzip_parse_count(), spz._device_gzip_reject_count()
                assert spz._compress_gzipped(data) == want
                assert spz._device_gzip_parse_count() == parses + 1
                assert spz._device_gzip_reject_count() == rejects


def test_default_route_at_the_baseline_size_equals_the_host_writer_and_zlib_inflates_it(tmp_path):
    """BASELINE configs[2]: 10 M points, SH degree 3 -> 650 MB of stream, the size the device route is the default for
    and was never driver-tested at.  member(device) == member(exact host writer, itself pinned to zlib by the CPU
    suite); zlib inflates it to the stream; loadSpz of the file with the device reader forced gives bit for bit what
    the stream decodes to."""
    n, deg = 10_000_000, 3
    g = cloud_of(n, deg, 3, antialiased=True)
    o = spz.PackOptions()
    o.from_coord = spz.RDF
    raw = spz._pack_to_stream(g, o)
    assert len(raw) == 16 + 65 * n
    with Env(SPZ_AMD_GZIP_DEVICE=None, SPZ_AMD_GZIP_VERIFY=None):   # the defaults
        parses, rejects = spz._device_gzip_parse_count(), spz._device_gzip_reject_count()
        member = spz._compress_gzipped(raw)
        assert spz._device_gzip_parse_count() == parses + 1, "the default route at this size is the device's"
        assert spz._device_gzip_reject_count() == rejects
    with Env(SPZ_AMD_GZIP_DEVICE="0"):
        host_member = spz._compress_gzipped(raw)
    assert member == host_member, "device writer and exact host writer disagree"
    del host_member
    assert zlib.decompress(member, 16 + 15) == raw, "zlib does not inflate the member to the stream"
    path = str(tmp_path / "cfg3.spz")
    with Env(SPZ_AMD_GZIP_DEVICE=None):
        assert spz.save_spz(g, o, path)
    with open(path, "rb") as f:
        assert f.read() == member, "saveSpz wrote something else than compressGzipped(stream)"
    del member
    u = spz.UnpackOptions()
    u.to_coord = spz.RDF
    with Env(SPZ_AMD_GUNZIP_DEVICE="1"):
        inflates = spz._device_inflate_count()
        back = spz.load_spz(path, u)
        assert spz._device_inflate_count() == inflates + 1, "the device reader declined the file"
    want = spz._unpack_from_stream(raw, u)
    assert same_cloud(back, want)


def test_three_callers_at_once_through_both_device_stages(tmp_path):
    """saveSpz + loadSpz of three different clouds from three threads, both container stages forced onto the device: the
    stream cache, the scratch slots, the kept-stream hand-over and the copy lane are all shared state (spz_host.cpp,
    spz_hostpath.hip, spz_lz77.hip).  Every file must be zlib's member of its own stream and load back to its own
    floats."""
    jobs = [(700_000, 3, 31), (1_000_000, 2, 32), (1_600_000, 1, 33)]
    clouds, raws, wants = {}, {}, {}
    o = spz.PackOptions()
    o.from_coord = spz.RUF
    u = spz.UnpackOptions()
    u.to_coord = spz.LUF
    for j in jobs:
        clouds[j] = cloud_of(j[0], j[1], j[2])
        raws[j] = spz._pack_to_stream(clouds[j], o)
        wants[j] = (zlib_gzip(raws[j]), spz._unpack_from_stream(raws[j], u))
    errors = []

    def work(j, reps):
        try:
            for r in range(reps):
                path = str(tmp_path / f"t{j[2]}_{r}.spz")
                if not spz.save_spz(clouds[j], o, path):
                    errors.append((j, r, "save failed"))
                    return
                with open(path, "rb") as f:
                    if f.read() != wants[j][0]:
                        errors.append((j, r, "file differs from zlib's member"))
                        return
                back = spz.load_spz(path, u)
                if not same_cloud(back, wants[j][1]):
                    errors.append((j, r, "loaded cloud differs"))
                    return
        except Exception as e:  # noqa: BLE001
            errors.append((j, "exception", repr(e)))

    with Env(SPZ_AMD_GZIP_DEVICE="1", SPZ_AMD_GUNZIP_DEVICE="1", SPZ_AMD_GZIP_VERIFY=None):
        parses, inflates, rejects = spz._device_gzip_parse_count(), spz._device_inflate_count(), spz._device_gzip_reject_count()
        threads = [threading.Thread(target=work, args=(j, 3)) for j in jobs]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        assert spz._device_gzip_reject_count() == rejects
        # a stage that finds the device's memory taken by the other callers may stand down (same bytes from the host);
        # most calls must have gone the device's way
        assert spz._device_gzip_parse_count() - parses >= 6, "the device writer mostly stood down"
        assert spz._device_inflate_count() - inflates >= 6, "the device reader mostly stood down"


def test_device_stages_stand_down_when_the_card_has_no_room():
    """SPZ_AMD_DEVICE_MEM_LIMIT_MIB makes the container stage believe the card is nearly full: writer and reader must
    decline before allocating anything and the host routes deliver the same bytes."""
    rng = np.random.default_rng(21)
    data = make("sh_like", 9_000_000, rng)
    want = zlib_gzip(data)
    with Env(SPZ_AMD_GZIP_DEVICE="1", SPZ_AMD_GUNZIP_DEVICE="1", SPZ_AMD_DEVICE_MEM_LIMIT_MIB="64"):
        parses, inflates = spz._device_gzip_parse_count(), spz._device_inflate_count()
        assert spz._compress_gzipped(data) == want
        assert spz._decompress_gzipped(want) == data
        assert spz._device_gzip_parse_count() == parses and spz._device_inflate_count() == inflates
    with Env(SPZ_AMD_GZIP_DEVICE="1", SPZ_AMD_GUNZIP_DEVICE="1", SPZ_AMD_DEVICE_MEM_LIMIT_MIB=None):
        assert spz._compress_gzipped(data) == want
        assert spz._decompress_gzipped(want) == data
        assert spz._device_gzip_parse_count() == parses + 1 and spz._device_inflate_count() == inflates + 1


def test_idle_scratch_is_bounded_by_the_keep_budget():
    """What the container stage leaves allocated on the card between calls is bounded by SPZ_AMD_SCRATCH_KEEP_MIB: with 0
    a call returns everything it took; by default the blocks stay for the next call (and go with
    spz_amd_release_device_memory)."""
    import torch
    from spz_amd import abi
    rng = np.random.default_rng(22)
    data = make("sh_like", 40_000_000, rng)      # ~23 bytes of scratch per input byte: ~0.9 GB
    L = abi.load_library()

    def free_now():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    def returned(before, slack=64 << 20):
        # the driver finishes a large free in the background: give it a moment
        import time
        for _ in range(40):
            if before - free_now() < slack:
                return True
            time.sleep(0.05)
        return False

    with Env(SPZ_AMD_GZIP_DEVICE="1", SPZ_AMD_SCRATCH_KEEP_MIB="0"):
        member = spz._compress_gzipped(data)      # first call: tables, streams, ... are made
        assert L.spz_amd_release_device_memory() == 0
        before = free_now()
        assert spz._compress_gzipped(data) == member
        assert returned(before), "scratch stayed allocated although the keep budget is 0"
    with Env(SPZ_AMD_GZIP_DEVICE="1", SPZ_AMD_SCRATCH_KEEP_MIB=None):
        before = free_now()
        assert spz._compress_gzipped(data) == member
        kept = before - free_now()
        assert kept > (400 << 20), "the default budget keeps the block for the next call"
        assert L.spz_amd_release_device_memory() == 0
        assert returned(before)
