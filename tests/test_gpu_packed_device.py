"""spz::loadSpzPackedDevice / spz.load_spz_packed_device (second half): loadSpzPacked
(load-spz.cc:609-632) with the packed sections left in device memory for a renderer.  The resident stream must be,
byte for byte, the stream the file holds; its sections must be deserializePackedGaussians' slices (:569-590); decoding
from it — all points, or an index list — must give the reference's floats (tests/golden/persplat.npz rows; the bulk
decode); and the raw pointers must be directly usable by the C ABI's device entry points."""
import os
import zlib

import numpy as np
import pytest

from conftest import FIELDS, assert_bits_equal, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def spz(cuda):
    import spz_amd.spz as m
    return m


def gz(b):
    co = zlib.compressobj(-1, zlib.DEFLATED, 16 + 15, 9, zlib.Z_DEFAULT_STRATEGY)
    return co.compress(b) + co.flush()


def device_bytes(ptr, nbytes, dev):
    from spz_amd.device import RawStream
    return RawStream(ptr, nbytes).tensor(dev).cpu().numpy()


def test_small_files_of_every_stream_version_against_the_per_splat_goldens(spz, cuda):
    """The reference's per-splat rows (v3 SH0..3, v2, v1 float16, fractionalBits 0/8/23) from a resident stream that the
    HOST readers inflated (these members are far below the device reader's size): sections, header fields, gather."""
    from spz_amd import abi
    g = load_golden("persplat.npz")
    cl, lg = load_golden("clouds.npz"), load_golden("legacy.npz")
    streams = {f"d{d}": cl[f"d{d}_stream_from0"] for d in range(4)}
    streams.update(v2=lg["v2_stream"], v1=lg["v1_stream"], fb0=lg["fb0_stream"], fb8=lg["fb8_stream"], fb23=lg["fb23_stream"])
    for name, s in streams.items():
        raw = s.tobytes()
        rc, hdr = abi.peek_header(raw)
        assert rc == 0
        d = spz._load_spz_packed_device_bytes(gz(raw))
        assert d.valid and not d.inflated_on_device
        assert (d.num_points, d.sh_degree, d.fractional_bits, d.version) == (hdr.num_points, hdr.sh_degree, hdr.fractional_bits, hdr.version)
        assert d.uses_float16 == (hdr.version == 1) and d.uses_quaternion_smallest_three == (hdr.version >= 3)
        assert d.stream_bytes == len(raw)
        assert device_bytes(d.stream_ptr, d.stream_bytes, cuda).tobytes() == raw
        lay = abi.stream_layout(hdr.num_points, hdr.sh_degree, hdr.version)
        order = ("positions", "alphas", "colors", "scales", "rotations", "sh")
        for k, sec in enumerate(order):
            ptr, nbytes = d.sections[sec]
            assert nbytes == lay.bytes[k]
            assert (ptr == 0) == (nbytes == 0)
            if nbytes:
                assert ptr == d.stream_ptr + lay.offset[k]
        idx = g[f"{name}_indices"]
        for to in (6, 1):
            o = spz.UnpackOptions()
            o.to_coord = spz.CoordinateSystem(to)
            c = d.unpack_indices([int(i) for i in idx], o)
            f = g[f"{name}_floats_4_{to}"]
            shd = {0: 0, 1: 3, 2: 8, 3: 15}[c.sh_degree]
            assert c.num_points == idx.size
            assert_bits_equal(c.positions, f[:, 0:3].reshape(-1), f"{name} positions")
            assert_bits_equal(c.rotations, f[:, 3:7].reshape(-1), f"{name} rotations")
            assert_bits_equal(c.scales, f[:, 7:10].reshape(-1), f"{name} scales")
            assert_bits_equal(c.colors, f[:, 10:13].reshape(-1), f"{name} colors")
            assert_bits_equal(c.alphas, f[:, 13], f"{name} alphas")
            sh = np.stack([f[:, 14:14 + shd], f[:, 29:29 + shd], f[:, 44:44 + shd]], axis=2)
            assert_bits_equal(c.sh, sh.reshape(-1), f"{name} sh")
            whole, want = d.unpack(o), spz._unpack_from_stream(raw, o)
            for k in FIELDS:
                assert_bits_equal(getattr(whole, k), getattr(want, k), f"{name} bulk {k}")
        d.release()
        assert not d.valid and d.stream_ptr == 0 and d.num_points == 0
        assert d.unpack(spz.UnpackOptions()).num_points == 0


def test_a_large_file_stays_where_the_device_reader_put_it(spz, cuda, tmp_path):
    """1.2 M points SH3 (78 MB of stream): the device reader inflates the member and the stream is never on the host.
    The resident bytes are the file's stream; a gather through the C++ layer, a gather and a bulk decode through the C
    ABI on the raw pointers all agree with the bulk decode of the stream."""
    import torch
    from spz_amd import abi, device as D
    from spz_amd.synth import make_cloud_numpy
    n, deg = 1_200_000, 3
    c = make_cloud_numpy(n, deg, 41)
    g = spz.GaussianCloud()
    g.sh_degree = deg
    for k in FIELDS:
        setattr(g, k, c[k])
    po = spz.PackOptions()
    po.from_coord = spz.RDF
    raw = spz._pack_to_stream(g, po)
    path = str(tmp_path / "big.spz")
    assert spz.save_spz(g, po, path)
    old = os.environ.get("SPZ_AMD_GUNZIP_DEVICE")
    os.environ["SPZ_AMD_GUNZIP_DEVICE"] = "1"
    try:
        d = spz.load_spz_packed_device(path)
    finally:
        if old is None:
            os.environ.pop("SPZ_AMD_GUNZIP_DEVICE", None)
        else:
            os.environ["SPZ_AMD_GUNZIP_DEVICE"] = old
    assert d.valid and d.inflated_on_device and d.num_points == n and d.sh_degree == deg and d.version == 3
    assert device_bytes(d.stream_ptr, d.stream_bytes, cuda).tobytes() == raw
    o = spz.UnpackOptions()
    o.to_coord = spz.LUF
    want = spz._unpack_from_stream(raw, o)
    rng = np.random.default_rng(4)
    idx = rng.integers(0, n, 5000, dtype=np.uint32)
    got = d.unpack_indices([int(i) for i in idx], o)
    per = {"positions": 3, "scales": 3, "rotations": 4, "alphas": 1, "colors": 3, "sh": 45}
    for k in FIELDS:
        rows = np.asarray(getattr(want, k)).reshape(n, per[k])[idx].reshape(-1)
        assert_bits_equal(getattr(got, k), rows, f"gather {k}")
    # the raw pointers in the C ABI's device entry points
    from spz_amd.device import RawStream
    st = RawStream(d.stream_ptr, d.stream_bytes).tensor(cuda)
    hdr = D.make_header(n, deg, 3, d.fractional_bits, d.antialiased)
    dev_idx = torch.from_numpy(idx.astype(np.int32)).to(cuda)
    gathered = D.decode_gather(st, hdr, dev_idx, abi.LUF)
    bulk = D.decode(st, hdr, abi.LUF)
    torch.cuda.synchronize()
    for k in FIELDS:
        rows = np.asarray(getattr(want, k)).reshape(n, per[k])[idx].reshape(-1)
        assert_bits_equal(gathered[k].cpu().numpy(), rows, f"ABI gather {k}")
        assert_bits_equal(bulk[k].cpu().numpy(), np.asarray(getattr(want, k)), f"ABI bulk {k}")
    whole = d.unpack(o)
    for k in FIELDS:
        assert_bits_equal(getattr(whole, k), getattr(want, k), f"unpack {k}")
    d.release()


def test_bad_files_give_the_empty_object(spz, capfd):
    """Failure results as loadSpzPacked's: not gzip -> empty silently (load-spz.cc:609-612); a stream with a wrong magic
    -> empty after the reference's log line."""
    assert not spz._load_spz_packed_device_bytes(b"not a gzip member at all").valid
    bad = bytearray(load_golden("clouds.npz")["d1_stream_from0"].tobytes())
    bad[0] ^= 0xff
    assert not spz._load_spz_packed_device_bytes(gz(bytes(bad))).valid
    assert not spz.load_spz_packed_device("/nonexistent/file.spz").valid
