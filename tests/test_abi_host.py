"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/spz_amd.h
declares, and its pure-host entry points (layout, header, tables, status strings) agree with the
oracle and the golden vectors.  No compute entry point is called with real work here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, assert_bits_equal, load_golden


@pytest.fixture(scope="module")
def lib():
    from spz_amd import abi
    return abi.load_library()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "spz_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spz_amd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib):
    from spz_amd import abi
    names = declared_functions()
    assert len(names) >= 17
    assert sorted(abi.EXPORTS) == names, "spz_amd/abi.py EXPORTS out of sync with include/spz_amd.h"
    for n in names:
        assert hasattr(lib, n), f"libspz_amd.so does not export {n}"
    assert lib.spz_amd_abi_version() == 1


def test_library_exports_nothing_but_the_declared_functions():
    """A drop-in library must not put unprefixed C names into a process: every function the dynamic symbol table
    defines is one include/spz_amd.h declares (C++ symbols are mangled under namespaces and cannot collide)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "spz_amd", "lib", "libspz_amd.so")],
                         check=True, capture_output=True, text=True).stdout
    declared = set(declared_functions())
    stray = []
    for line in out.splitlines():
        parts = line.split()
        if len(parts) != 3 or parts[1] not in "TtWw":
            continue
        name = parts[2]
        if name.startswith("_Z") or name in ("_init", "_fini"):
            continue
        if name not in declared:
            stray.append(name)
    assert not stray, f"libspz_amd.so exports undeclared C symbols: {stray}"


def test_host_library_and_python_module_load():
    host = C.CDLL(os.path.join(ROOT, "spz_amd", "lib", "libspz_host.so"))
    assert host is not None
    import spz_amd.spz as spz
    for name in ("GaussianCloud", "PackOptions", "UnpackOptions", "CoordinateSystem", "load_spz", "save_spz",
                 "load_splat_from_ply", "save_splat_to_ply", "UNSPECIFIED", "LDB", "RDB", "LUB", "RUB", "LDF",
                 "RDF", "LUF", "RUF"):
        assert hasattr(spz, name), name


def test_status_strings(lib):
    from spz_amd import abi
    assert abi.status_string(abi.OK) == "ok"
    assert abi.status_string(abi.ERR_HEADER_NOT_FOUND) == "header not found"
    assert abi.status_string(abi.ERR_SHORT_STREAM) == "read error"
    assert abi.status_string(-999) == "unknown status"


@pytest.mark.parametrize("version", [1, 2, 3])
@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_stream_layout_matches_oracle(lib, oracle, deg, version):
    from spz_amd import abi
    for n in (0, 1, 7, 4096, 10_000_000, 80_000_000):
        lay = abi.stream_layout(n, deg, version)
        assert lay.total_bytes == oracle.lib.spzo_stream_size(n, deg, version)
        pos_b = 6 if version == 1 else 9
        rot_b = 4 if version == 3 else 3
        d = {0: 0, 1: 9, 2: 24, 3: 45}[deg]
        assert list(lay.bytes_per_point) == [pos_b, 1, 3, 3, rot_b, d]
        off = 16
        for s in range(6):
            assert lay.offset[s] == off and lay.bytes[s] == n * lay.bytes_per_point[s]
            off += lay.bytes[s]
    bad = abi.Layout()
    assert lib.spz_amd_stream_layout(1, 4, 3, C.byref(bad)) == abi.ERR_INVALID_ARG
    assert lib.spz_amd_stream_layout(1, 3, 4, C.byref(bad)) == abi.ERR_INVALID_ARG
    assert lib.spz_amd_stream_layout(1, 3, 0, C.byref(bad)) == abi.ERR_INVALID_ARG


def test_header_write_and_peek_round_trip(lib, oracle):
    from spz_amd import abi
    g = load_golden("kat_small.npz")
    # the header the reference wrote for the 2-point cloud
    assert abi.write_header(3, 2, 3, 12, True) == g["two_stream_from0"][:16].tobytes()
    assert abi.write_header(3, 0, 0, 12, False) == g["empty_stream"][:16].tobytes()
    for name in ("two_stream_from0", "empty_stream", "shedge_stream", "two_sh0_stream"):
        s = g[name]
        rc, h = abi.peek_header(s.tobytes())
        orc, oh = oracle.peek(s)
        assert rc == 0 and orc == 0
        assert (h.version, h.num_points, h.sh_degree, h.fractional_bits, h.antialiased) == \
               (oh["version"], oh["num_points"], oh["sh_degree"], oh["fractional_bits"], oh["antialiased"])


def test_peek_header_rejections_mirror_the_reference(lib, oracle):
    """deserializePackedGaussians load-spz.cc:553-568,591-594; golden: the reference returned an
    empty cloud for each of these streams."""
    from spz_amd import abi
    g = load_golden("legacy.npz")
    want = {"magic": abi.ERR_HEADER_NOT_FOUND, "tiny": abi.ERR_HEADER_NOT_FOUND, "version4": abi.ERR_VERSION,
            "version0": abi.ERR_VERSION, "toomany": abi.ERR_TOO_MANY_POINTS, "shdeg4": abi.ERR_SH_DEGREE,
            "short": abi.ERR_SHORT_STREAM}
    oracle_codes = {abi.ERR_HEADER_NOT_FOUND: -1, abi.ERR_VERSION: -2, abi.ERR_TOO_MANY_POINTS: -3,
                    abi.ERR_SH_DEGREE: -4, abi.ERR_SHORT_STREAM: -5}
    for name, code in want.items():
        s = g[f"bad_{name}_stream"]
        assert int(g[f"bad_{name}_numpoints"]) == 0
        rc, h = abi.peek_header(s.tobytes())
        assert rc == code and h is None, name
        assert oracle.peek(s)[0] == oracle_codes[code], name
    # exactly 10 M points passes the reference's limit, 10 M + 1 does not (load-spz.cc:549,561)
    hdr = abi.write_header(3, 10_000_000, 0)
    assert abi.peek_header(hdr)[0] == abi.ERR_SHORT_STREAM
    assert abi.peek_header(abi.write_header(3, 10_000_001, 0))[0] == abi.ERR_TOO_MANY_POINTS
    # the _ex form lifts the limit for reassembled shard streams
    assert abi.peek_header(abi.write_header(3, 80_000_000, 3), max_points=0)[0] == abi.ERR_SHORT_STREAM
    assert abi.peek_header(b"")[0] == abi.ERR_HEADER_NOT_FOUND
    for ver in ("v1", "v2"):
        rc, h = abi.peek_header(g[f"{ver}_stream"].tobytes())
        assert rc == 0 and h.version == int(ver[1])


def test_tables_equal_reference_tables(lib):
    """Host-computed decode tables / alpha thresholds == the reference's (tests/golden/tables.npz)."""
    from spz_amd import abi
    g = load_golden("tables.npz")
    a, c, t = abi.get_tables()
    assert_bits_equal(a, g["alpha_decode"], "alpha decode table")
    assert_bits_equal(c, g["color_decode"], "colour decode table")
    assert_bits_equal(t, g["alpha_thresholds"], "alpha thresholds")


def test_compute_entry_points_fail_loudly_without_a_gpu(lib):
    """No CPU fallback: with no HIP device the host-pointer entry points return ERR_NO_DEVICE."""
    import torch
    from spz_amd import abi
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    n = 4
    arrs = [np.zeros(m, np.float32) for m in (3 * n, 3 * n, 4 * n, n, 3 * n, 0)]
    arrs[2][3::4] = 1
    p = abi.CloudPtrs(*[a.ctypes.data if a.size else None for a in arrs])
    out = np.zeros(abi.stream_layout(n, 0, 3).total_bytes, np.uint8)
    rc = lib.spz_amd_encode_host(C.byref(p), n, 0, 0, 0, 3, out.ctypes.data, out.size, 0)
    assert rc == abi.ERR_NO_DEVICE
    assert not out.any(), "nothing may be written without a device"
    stream = np.frombuffer(abi.write_header(3, n, 0), np.uint8)
    stream = np.concatenate([stream, np.zeros(20 * n, np.uint8)])
    rc = lib.spz_amd_decode_host(stream.ctypes.data, stream.size, 0, C.byref(p), 0)
    assert rc == abi.ERR_NO_DEVICE
    assert lib.spz_amd_device_count() == 0


def test_ten_million_point_cap_sits_where_the_reference_has_it(lib):
    """The 10 M limit belongs to deserializePackedGaussians (load-spz.cc:549,561): spz_amd_decode_host (the
    loadSpz route) applies it, spz_amd_decode_host_ex(max_points=0) (the unpackGaussians route, :467-531,
    which has no limit) does not.  Status codes only: a size-consistent 11 M-point SH0 header."""
    import torch
    from spz_amd import abi
    n = 11_000_000
    size = abi.stream_layout(n, 0, 3).total_bytes
    stream = np.zeros(size, np.uint8)             # untouched pages: costs nothing until read
    stream[:16] = np.frombuffer(abi.write_header(3, n, 0), np.uint8)
    out = [np.zeros(1, np.float32) for _ in range(5)]
    p = abi.CloudPtrs(*[a.ctypes.data for a in out], None)
    assert lib.spz_amd_decode_host(stream.ctypes.data, size, 0, C.byref(p), 0) == abi.ERR_TOO_MANY_POINTS
    assert lib.spz_amd_decode_host_ex(stream.ctypes.data, size, 10_999_999, 0, C.byref(p), 0) == abi.ERR_TOO_MANY_POINTS
    if not torch.cuda.is_available():
        # past the header checks and the argument checks; the device is what is missing
        assert lib.spz_amd_decode_host_ex(stream.ctypes.data, size, 0, 0, C.byref(p), 0) == abi.ERR_NO_DEVICE
        assert lib.spz_amd_decode_host_ex(stream.ctypes.data, size - 1, 0, 0, C.byref(p), 0) == abi.ERR_SHORT_STREAM


def test_shard_fragments_match_the_python_plan(lib):
    """spz_amd_shard_fragments (the table spz_amd_gatherv_rccl walks) against spz_amd.shard.ShardPlan.fragments
    (what the torch route and the gloo tests use)."""
    from spz_amd import abi, shard
    for n, deg, ver, world in ((1000, 3, 3, 3), (80_000_000, 3, 3, 8), (17, 0, 2, 4), (5, 1, 3, 8)):
        plan = shard.plan_even(n, deg, world, ver)
        for r in range(world):
            f = abi.Fragments()
            assert lib.spz_amd_shard_fragments(plan.first[r], plan.count[r], n, deg, ver, C.byref(f)) == abi.OK
            assert [(f.global_offset[s], f.local_offset[s], f.bytes[s]) for s in range(6)] == plan.fragments(r)
    f = abi.Fragments()
    assert lib.spz_amd_shard_fragments(5, 6, 10, 0, 3, C.byref(f)) == abi.ERR_INVALID_ARG
    assert lib.spz_amd_shard_fragments(0, 1, 1, 4, 3, C.byref(f)) == abi.ERR_INVALID_ARG
    # the exchange entry points validate before they touch RCCL or the device
    assert lib.spz_amd_gatherv_rccl(None, 0, 1, 0, None, None, 0, 3, None, None, 0x3f, None) == abi.ERR_INVALID_ARG
    assert lib.spz_amd_ipc_open(None, None) == abi.ERR_INVALID_ARG
    assert lib.spz_amd_encode_shard_sections_device(None, 0, 0, 0, 0, 0, 0, 3, 0, 0x40, None, 0, None) == abi.ERR_INVALID_ARG


def test_python_module_raises_without_a_gpu():
    import torch
    import spz_amd.spz as spz
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    c = spz.GaussianCloud()
    c.positions = np.zeros(3, np.float32)
    c.scales = np.zeros(3, np.float32)
    c.rotations = np.array([0, 0, 0, 1], np.float32)
    c.alphas = np.zeros(1, np.float32)
    c.colors = np.zeros(3, np.float32)
    with pytest.raises(RuntimeError, match="no usable HIP device"):
        spz.save_spz(c, spz.PackOptions(), os.path.join("/tmp", "spz_amd_never_written.spz"))


def test_argument_validation_happens_before_any_device_work(lib):
    """Every argument error is reported as such even on a machine without a GPU (validation comes
    first, ERR_NO_DEVICE last), with the reference's checkSizes-style conditions (load-spz.cc:106-127)."""
    from spz_amd import abi
    n = 8
    arrs = [np.zeros(m, np.float32) for m in (3 * n, 3 * n, 4 * n, n, 3 * n, 9 * n)]
    good = abi.CloudPtrs(*[a.ctypes.data for a in arrs])
    no_sh = abi.CloudPtrs(*[a.ctypes.data for a in arrs[:5]], None)
    out = np.zeros(abi.stream_layout(n, 1, 3).total_bytes, np.uint8)
    enc = lambda cloud, deg=1, frm=0, ver=3, cap=out.size, dst=out.ctypes.data: lib.spz_amd_encode_device(
        C.byref(cloud) if cloud is not None else None, n, deg, 0, frm, ver, dst, cap, None)
    assert enc(None) == abi.ERR_INVALID_ARG
    assert enc(good, dst=None) == abi.ERR_INVALID_ARG
    assert enc(good, frm=9) == abi.ERR_INVALID_ARG
    assert enc(good, frm=-1) == abi.ERR_INVALID_ARG
    assert enc(good, deg=4) == abi.ERR_INVALID_ARG
    assert enc(good, ver=4) == abi.ERR_INVALID_ARG
    assert enc(good, ver=1) == abi.ERR_UNSUPPORTED
    assert enc(good, cap=out.size - 1) == abi.ERR_CAPACITY
    assert enc(no_sh) == abi.ERR_INVALID_ARG          # sh missing although sh_degree = 1
    hdr = abi.Header(3, n, 1, 12, 0, 0)
    dec = lambda h, size=out.size, to=0, cloud=good, src=out.ctypes.data: lib.spz_amd_decode_device(
        src, size, C.byref(h) if h is not None else None, to, C.byref(cloud) if cloud is not None else None, None)
    assert dec(None) == abi.ERR_INVALID_ARG
    assert dec(hdr, src=None) == abi.ERR_INVALID_ARG
    assert dec(hdr, cloud=None) == abi.ERR_INVALID_ARG
    assert dec(hdr, to=9) == abi.ERR_INVALID_ARG
    assert dec(abi.Header(0, n, 1, 12, 0, 0)) == abi.ERR_VERSION
    assert dec(abi.Header(3, n, 4, 12, 0, 0)) == abi.ERR_SH_DEGREE
    assert dec(hdr, size=out.size - 1) == abi.ERR_SHORT_STREAM
    assert dec(hdr, cloud=no_sh) == abi.ERR_INVALID_ARG
    # shards must lie inside the stream
    assert lib.spz_amd_encode_shard_device(C.byref(good), 4, 8, 8, 1, 0, 0, 3, 0, out.ctypes.data, out.size, None) \
        == abi.ERR_INVALID_ARG
    assert lib.spz_amd_decode_shard_device(out.ctypes.data, out.size, C.byref(hdr), 9, 1, 0, C.byref(good), None) \
        == abi.ERR_INVALID_ARG
    # zero points: nothing to do, no device needed
    assert lib.spz_amd_convert_coordinates_device(None, None, None, 0, 0, 4, 6, None) == abi.OK
    assert lib.spz_amd_decode_shard_device(out.ctypes.data, out.size, C.byref(hdr), 3, 0, 0, C.byref(good), None) == abi.OK
    # .ply column maps
    cols = abi.PlyColumns()
    assert lib.spz_amd_ply_default_columns(3, C.byref(cols)) == abi.OK and cols.stride == 26
    bad = abi.PlyColumns.from_buffer_copy(bytes(cols))
    bad.alpha = 26
    assert lib.spz_amd_ply_rows_to_cloud_device(out.ctypes.data, n, C.byref(bad), 0, C.byref(good), None) == abi.ERR_INVALID_ARG
    bad = abi.PlyColumns.from_buffer_copy(bytes(cols))
    bad.stride = 300
    assert lib.spz_amd_ply_rows_to_cloud_device(out.ctypes.data, n, C.byref(bad), 0, C.byref(good), None) == abi.ERR_INVALID_ARG
    assert lib.spz_amd_cloud_to_ply_rows_device(C.byref(good), n, 16, 0, out.ctypes.data, None) == abi.ERR_INVALID_ARG
    # host gather: header checks of the stream, then arguments, then the device
    hs = np.zeros(out.size, np.uint8)
    assert lib.spz_amd_write_header(C.byref(hdr), hs.ctypes.data) == abi.OK
    idx = np.array([0, 3], np.uint32)
    gat = lambda stream=hs.ctypes.data, size=hs.size, ind=idx.ctypes.data, cnt=2, to=0, cloud=good: \
        lib.spz_amd_decode_gather_host(stream, size, 0, ind, cnt, to, C.byref(cloud) if cloud is not None else None, 0)
    assert gat(stream=None) == abi.ERR_INVALID_ARG
    assert gat(cloud=None) == abi.ERR_INVALID_ARG
    assert gat(to=12) == abi.ERR_INVALID_ARG
    assert gat(size=10) == abi.ERR_HEADER_NOT_FOUND
    assert gat(size=hs.size - 1) == abi.ERR_SHORT_STREAM
    assert gat(cnt=0) == abi.OK
    assert gat(ind=None) == abi.ERR_INVALID_ARG
    assert gat(cloud=no_sh) == abi.ERR_INVALID_ARG
    # an index past the end is an argument error of the host form (the device form clamps, as documented)
    bad_idx = np.array([0, n], np.uint32)
    assert gat(ind=bad_idx.ctypes.data) == abi.ERR_INVALID_ARG
    # median selection: arguments first
    one = np.zeros(3, np.float32)
    res = np.zeros(1, np.float32)
    assert lib.spz_amd_median_scale_sum_host(None, 1, res.ctypes.data, 0) == abi.ERR_INVALID_ARG
    assert lib.spz_amd_median_scale_sum_host(one.ctypes.data, 0, res.ctypes.data, 0) == abi.ERR_INVALID_ARG
    assert lib.spz_amd_median_scale_sum_host(one.ctypes.data, 1, None, 0) == abi.ERR_INVALID_ARG
    assert lib.spz_amd_median_scale_sum_device(one.ctypes.data, 1, None, res.ctypes.data, None) == abi.ERR_INVALID_ARG
    assert lib.spz_amd_median_scale_sum_device(one.ctypes.data, 1 << 32, res.ctypes.data, res.ctypes.data, None) \
        == abi.ERR_INVALID_ARG
    assert lib.spz_amd_release_device_memory() == abi.OK   # harmless without a device
