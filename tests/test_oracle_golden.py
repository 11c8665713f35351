"""Pins the CPU restatement (oracle/spz_oracle.c) to the golden vectors produced by the
reference's own C++ (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import FIELDS, assert_bits_equal, assert_bytes_equal, load_golden


def cloud_from(g, prefix):
    return {k: g[f"{prefix}_{k}"] for k in FIELDS}


def test_two_point_cloud_every_from(oracle):
    g = load_golden("kat_small.npz")
    c = cloud_from(g, "two_in")
    for frm in range(9):
        assert_bytes_equal(oracle.pack(c, 2, 3, True, frm), g[f"two_stream_from{frm}"], f"from={frm}")


def test_two_point_cloud_appendix_c_bytes(oracle):
    """SURVEY.md Appendix C known-answer bytes (measured on the compiled reference)."""
    g = load_golden("kat_small.npz")
    c = cloud_from(g, "two_in")
    want = {0: ("0000009a0100cdfcffcd0400660600000800", "7df691b330575ec6", "000008080810101818"),
            6: ("00000066feff330300cd04009af9ff00f8ff", "7df49193305556c6", "fffff8f8f8f0101818"),
            7: ("0000009a010033030033fbff66060000f8ff", "7df499b330555ee6", "000008f8f8f0f0f0e8")}
    for frm, (pos, rot, sh) in want.items():
        s = oracle.pack(c, 2, 3, True, frm)
        assert s[16:34].tobytes().hex() == pos
        assert s[16 + 32:16 + 40].tobytes().hex() == rot
        assert s[16 + 40:16 + 49].tobytes().hex() == sh
        assert s[16 + 18:16 + 20].tobytes().hex() == "45ba"                # alphas
        assert s[16 + 20:16 + 26].tobytes().hex() == "5980a66c9383"        # colours
        assert s[16 + 26:16 + 32].tobytes().hex() == "70808890a0a2"        # scales
    assert s[:16].tobytes().hex() == "4e47535003000000020000000" + "30c0100"


def test_two_point_cloud_every_to(oracle):
    g = load_golden("kat_small.npz")
    s = g["two_stream_from0"]
    for to in range(9):
        rc, u = oracle.unpack(s, to)
        assert rc == 0 and u["num_points"] == 2 and u["sh_degree"] == 3 and u["antialiased"]
        for k in FIELDS:
            assert_bits_equal(u[k], g[f"two_dec_to{to}_{k}"], f"to={to} {k}")


def test_sh_edge_kat(oracle):
    """reference tests/python/load_spz_test.py:180-207"""
    g = load_golden("kat_small.npz")
    c = cloud_from(g, "shedge_in")
    s = oracle.pack(c, 1, 1, False, 0)
    assert_bytes_equal(s, g["shedge_stream"])
    rc, u = oracle.unpack(s, 0)
    assert rc == 0
    np.testing.assert_allclose(u["sh"], [0.0, 0.0, 0.0, -1.0, -1.0, -0.9375, 0.9375, 0.9922, 0.9922], atol=2e-5)
    for k in FIELDS:
        assert_bits_equal(u[k], g[f"shedge_dec_{k}"], k)


def test_coordinate_kat(oracle):
    """reference tests/python/load_spz_test.py:444-512"""
    g = load_golden("kat_small.npz")
    c = cloud_from(g, "coord_in")
    s = oracle.pack(c, 1, 1, False, 4)
    assert_bytes_equal(s, g["coord_stream_from4"])
    rc, u = oracle.unpack(s, 6)
    np.testing.assert_allclose(u["positions"], [1.0, -2.0, -3.0], atol=1 / 2048.0)
    for k in FIELDS:
        assert_bits_equal(u[k], g[f"coord_dec_from4_to6_{k}"], k)
    c0 = dict(c, sh=np.zeros(0, np.float32))
    s = oracle.pack(c0, 1, 0, False, 6)
    assert_bytes_equal(s, g["coord_stream_sh0_from6"])
    rc, u = oracle.unpack(s, 7)
    np.testing.assert_allclose(u["positions"], [-1.0, -2.0, 3.0], atol=1 / 2048.0)
    for k in FIELDS:
        assert_bits_equal(u[k], g[f"coord_dec_from6_to7_{k}"], k)


def test_empty_cloud(oracle):
    g = load_golden("kat_small.npz")
    e = {k: np.zeros(0, np.float32) for k in FIELDS}
    s = oracle.pack(e, 0, 0, False, 0)
    assert_bytes_equal(s, g["empty_stream"])
    rc, u = oracle.unpack(s, 0)
    assert rc == 0 and u["num_points"] == 0


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_seeded_clouds(oracle, deg):
    g = load_golden("clouds.npz")
    c = cloud_from(g, f"d{deg}_in")
    n = c["alphas"].size
    for frm in (0, 6, 7):
        assert_bytes_equal(oracle.pack(c, n, deg, bool(deg & 1), frm), g[f"d{deg}_stream_from{frm}"],
                           f"deg={deg} from={frm}")
    for to in (0, 1, 6, 7):
        rc, u = oracle.unpack(g[f"d{deg}_stream_from0"], to)
        assert rc == 0
        for k in FIELDS:
            assert_bits_equal(u[k], g[f"d{deg}_dec_to{to}_{k}"], f"deg={deg} to={to} {k}")


def test_odd_sizes(oracle):
    g = load_golden("clouds.npz")
    names = sorted({k.split("_in_")[0] for k in g.files if k.startswith("odd_") and "_in_" in k})
    assert len(names) >= 40
    for nm in names:
        n = int(nm.split("_")[1][1:])
        deg = int(nm.split("_")[2][1:])
        c = cloud_from(g, f"{nm}_in")
        s = oracle.pack(c, n, deg, False, 6)
        assert_bytes_equal(s, g[f"{nm}_stream_from6"], nm)
        rc, u = oracle.unpack(s, 7)
        for k in FIELDS:
            assert_bits_equal(u[k], g[f"{nm}_dec_to7_{k}"], f"{nm} {k}")


def test_legacy_versions_and_fractional_bits(oracle):
    g = load_golden("legacy.npz")
    for ver, tos in (("v2", (0, 6, 7)), ("v1", (0, 6))):
        for to in tos:
            rc, u = oracle.unpack(g[f"{ver}_stream"], to)
            assert rc == 0
            for k in FIELDS:
                assert_bits_equal(u[k], g[f"{ver}_dec_to{to}_{k}"], f"{ver} to={to} {k}")
    for fb in (0, 8, 16, 23):
        rc, u = oracle.unpack(g[f"fb{fb}_stream"], 6)
        assert_bits_equal(u["positions"], g[f"fb{fb}_dec_positions"], f"fb={fb}")


def test_rejected_headers(oracle):
    g = load_golden("legacy.npz")
    want = {"magic": -1, "version4": -2, "version0": -2, "toomany": -3, "shdeg4": -4, "short": -5, "tiny": -1}
    for name, rc_want in want.items():
        assert int(g[f"bad_{name}_numpoints"]) == 0  # the reference returned an empty cloud
        rc, _ = oracle.peek(g[f"bad_{name}_stream"])
        assert rc == rc_want, name


def test_decode_tables(oracle):
    g = load_golden("tables.npz")
    assert_bits_equal(oracle.alpha_decode_table(), g["alpha_decode"], "alpha")
    assert_bits_equal(oracle.color_decode_table(), g["color_decode"], "colour")
    a = g["alpha_decode"]
    assert a[0] == -np.inf and a[255] == np.inf


def test_alpha_thresholds_step_exactly(oracle):
    g = load_golden("tables.npz")
    t = g["alpha_thresholds"]
    assert t.size == 255 and np.all(np.diff(t) > 0)
    below = np.nextafter(t, np.float32(-np.inf), dtype=np.float32)
    for v in range(1, 256):
        assert oracle.alpha_byte(t[v - 1]) >= v
        assert oracle.alpha_byte(below[v - 1]) < v


def test_quaternion_golden_sets(oracle):
    """Per-quaternion helpers against tests/golden/quats.npz (edge / near-tie encodes, arbitrary
    32-bit decodes incl. sum-of-squares > 1 -> NaN, every v2 byte value)."""
    import ctypes as C
    g = load_golden("quats.npz")
    L = oracle.lib
    L.spzo_coordinate_converter.argtypes = [C.c_int, C.c_int, C.c_void_p]
    for fn in ("spzo_pack_quat_smallest_three", "spzo_unpack_quat_smallest_three", "spzo_unpack_quat_first_three"):
        getattr(L, fn).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        getattr(L, fn).restype = None
    conv = (C.c_float * 21)()
    q = np.ascontiguousarray(g["enc_in"], np.float32).reshape(-1, 4)
    for frm in (0, 6, 7, 1):
        L.spzo_coordinate_converter(frm, 4, conv)
        got = np.zeros(q.shape[0] * 4, np.uint8)
        for i in range(q.shape[0]):
            L.spzo_pack_quat_smallest_three(got[4 * i:].ctypes.data, q[i].ctypes.data, conv)
        assert_bytes_equal(got, g[f"enc_bytes_from{frm}"], f"encode from={frm}")
    rb = np.ascontiguousarray(g["dec3_bytes"])
    r2 = np.ascontiguousarray(g["dec2_bytes"])
    for to in (0, 6, 7):
        L.spzo_coordinate_converter(4, to, conv)
        out = np.zeros(rb.size, np.float32)
        for i in range(rb.size // 4):
            L.spzo_unpack_quat_smallest_three(out[4 * i:].ctypes.data, rb[4 * i:].ctypes.data, conv)
        assert_bits_equal(out, g[f"dec3_to{to}"], f"v3 decode to={to}")
        out = np.zeros(r2.size // 3 * 4, np.float32)
        for i in range(r2.size // 3):
            L.spzo_unpack_quat_first_three(out[4 * i:].ctypes.data, r2[3 * i:].ctypes.data, conv)
        assert_bits_equal(out, g[f"dec2_to{to}"], f"v2 decode to={to}")
    assert np.isnan(g["dec3_to0"]).any(), "the set must exercise the NaN (sum > 1) branch"


def test_median_volume_golden(oracle):
    """spzo_median_volume == the reference's GaussianCloud::medianVolume (splat-types.h:170-185) on odd and
    even counts, ties, signed zeros, infinities and a wide dynamic range, to the bit."""
    g = load_golden("median.npz")
    names = sorted(k[:-7] for k in g.files if k.endswith("_scales"))
    assert len(names) >= 15
    for name in names:
        sc = g[f"{name}_scales"]
        got = oracle.median_volume(sc, sc.size // 3)
        assert np.float32(got).tobytes() == np.float32(g[f"{name}_volume"]).tobytes(), name
    assert oracle.median_volume(np.zeros(0, np.float32), 0) == np.float32(0.01)
