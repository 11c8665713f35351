"""Host-side behaviour of the Python module `spz_amd.spz` that needs no GPU: the enum, the option
structs, GaussianCloud's properties and their validation.  Restates, with the same values and
error texts, what the reference's suite pins for its nanobind shim
(/root/reference/tests/python/load_spz_test.py; src/python/spz/spz.cc) — line numbers per test.
Anything that quantises, dequantises or flips goes through the device and lives in
test_gpu_python_module.py."""
import os

import numpy as np
import pytest

import spz_amd.spz as spz

NAMES = ["UNSPECIFIED", "LDB", "RDB", "LUB", "RUB", "LDF", "RDF", "LUF", "RUF"]


def test_reference_module_name_imports():
    """`import spz` (the reference's package name, src/python/spz/__init__.py:1-2) resolves to this implementation with
    the shim's public names."""
    import importlib
    from conftest import ROOT
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    ref_name = importlib.import_module("spz")
    for k in ("GaussianCloud", "PackOptions", "UnpackOptions", "CoordinateSystem", "load_spz", "save_spz",
              "load_splat_from_ply", "save_splat_to_ply", "RUB", "RDF"):
        assert getattr(ref_name, k) is getattr(spz, k), k
    assert not any(n.startswith("_") for n in ref_name.__all__)


def test_coordinate_system_enum_values():
    """load_spz_test.py:233-261; numeric order of splat-types.h:33-43."""
    assert hasattr(spz, "CoordinateSystem")
    values = [getattr(spz, k) for k in NAMES]
    assert len(set(values)) == 9
    assert [int(v) for v in values] == list(range(9))
    for k in NAMES:
        assert getattr(spz.CoordinateSystem, k) == getattr(spz, k)


def test_options_default_to_unspecified_and_are_mutable():
    """load_spz_test.py:264-291."""
    p, u = spz.PackOptions(), spz.UnpackOptions()
    assert p.from_coord == spz.UNSPECIFIED and u.to_coord == spz.UNSPECIFIED
    for k in NAMES:
        p.from_coord = getattr(spz, k)
        u.to_coord = getattr(spz, k)
        assert p.from_coord == getattr(spz, k) and u.to_coord == getattr(spz, k)


def test_new_cloud_is_empty():
    """load_spz_test.py:294-309."""
    c = spz.GaussianCloud()
    assert c.num_points == 0 and c.sh_degree == 0 and c.antialiased is False
    for k in ("positions", "scales", "rotations", "alphas", "colors", "sh"):
        a = getattr(c, k)
        assert isinstance(a, np.ndarray) and a.dtype == np.float32 and len(a) == 0


def test_property_setting_and_read_only_num_points():
    """load_spz_test.py:312-351."""
    c = spz.GaussianCloud()
    with pytest.raises(AttributeError):
        c.num_points = 5
    c.sh_degree = 2
    c.antialiased = True
    assert c.sh_degree == 2 and c.antialiased is True
    c.positions = np.array([1.0, 2.0, 3.0], np.float32)
    assert c.num_points == 1
    vals = {
        "scales": np.array([0.1, 0.2, 0.3], np.float32),
        "rotations": np.array([0.0, 0.0, 0.0, 1.0], np.float32),
        "alphas": np.array([0.5], np.float32),
        "colors": np.array([1.0, 0.0, 0.0], np.float32),
        "sh": np.zeros(24, np.float32),
    }
    for k, v in vals.items():
        setattr(c, k, v)
        got = getattr(c, k)
        assert got.dtype == np.float32
        np.testing.assert_array_equal(got, v)
    np.testing.assert_array_equal(c.positions, [1.0, 2.0, 3.0])


def test_property_arrays_are_copies():
    """The shim copies in both directions (spz.cc: vector<float> <-> ndarray): later edits of the
    source array or of a returned array do not reach the cloud."""
    c = spz.GaussianCloud()
    src = np.array([1.0, 2.0, 3.0], np.float32)
    c.positions = src
    src[0] = 99.0
    got = c.positions
    assert got[0] == 1.0
    got[1] = -7.0
    assert c.positions[1] == 2.0


def test_array_dtype_handling():
    """load_spz_test.py:354-372: numeric dtypes convert to float32, strings and complex are rejected
    with the binding layer's 'incompatible function arguments' TypeError."""
    c = spz.GaussianCloud()
    for dt in (np.float64, np.int32, np.float32, np.int64, np.uint8, np.float16):
        c.positions = np.array([1, 2, 3], dtype=dt)
        assert c.positions.dtype == np.float32
        np.testing.assert_array_equal(c.positions, [1.0, 2.0, 3.0])
    with pytest.raises(TypeError, match="incompatible function arguments"):
        c.positions = np.array(["a", "b", "c"], dtype=np.str_)
    with pytest.raises(TypeError, match="incompatible function arguments"):
        c.positions = np.array([1 + 2j, 3 + 4j, 5 + 6j], dtype=np.complex64)
    np.testing.assert_array_equal(c.positions, [1.0, 2.0, 3.0])   # a rejected assignment changes nothing


def test_non_contiguous_and_2d_inputs():
    """A strided view is accepted and read in logical order."""
    c = spz.GaussianCloud()
    base = np.arange(12, dtype=np.float32)
    c.positions = base[::2]
    np.testing.assert_array_equal(c.positions, base[::2])
    assert c.num_points == 2


def test_median_volume_of_an_empty_cloud():
    """load_spz_test.py:403-408 and splat-types.h:171-173: 0.01 for an empty cloud, no device involved
    (the selection over a non-empty cloud runs on the GPU: test_gpu_python_module.py)."""
    c = spz.GaussianCloud()
    assert abs(c.median_volume() - 0.01) < 1e-6


def test_empty_arrays_are_accepted():
    """load_spz_test.py:677-695."""
    c = spz.GaussianCloud()
    e = np.array([], np.float32)
    for k in ("positions", "scales", "rotations", "alphas", "colors", "sh"):
        setattr(c, k, e)
        assert len(getattr(c, k)) == 0
    assert c.num_points == 0


def test_shape_validation_messages():
    """load_spz_test.py:810-839: the setters' ValueError texts (spz.cc property setters)."""
    c = spz.GaussianCloud()
    with pytest.raises(TypeError):
        c.positions = np.array([1 + 2j, 3 + 4j], dtype=np.complex64)
    c.sh_degree = 2
    c.positions = np.zeros(3, np.float32)
    with pytest.raises(ValueError, match="positions length must be a multiple of 3"):
        c.positions = np.array([1, 2], np.float32)
    with pytest.raises(ValueError, match=r"scales length must equal num_points \* 3"):
        c.scales = np.zeros(6, np.float32)
    with pytest.raises(ValueError, match=r"rotations length must equal num_points \* 4"):
        c.rotations = np.zeros(8, np.float32)
    with pytest.raises(ValueError, match=r"colors length must equal num_points \* 3"):
        c.colors = np.zeros(6, np.float32)
    with pytest.raises(ValueError, match="sh must be empty when sh_degree == 0"):
        c.sh_degree = 0
        c.sh = np.zeros(3, np.float32)
    c.sh_degree = 2
    with pytest.raises(ValueError, match="sh length must be a multiple of 24, got 45"):
        c.sh = np.zeros(45, np.float32)
    with pytest.raises(ValueError, match=r"sh length must equal num_points \* \(\(sh_degree\+1\)\^2 - 1\) \* 3"):
        c.sh = np.zeros(48, np.float32)
    # shim quirk kept as is (spz.cc:226-230): the count check runs AFTER the assignment, so the rejected
    # 6-element scales array is what the cloud now holds; the multiple-of-k check runs before it.
    assert c.num_points == 1 and len(c.scales) == 6 and len(c.sh) == 48


def test_file_error_conventions_need_no_device(tmp_path):
    """load_spz_test.py:842-863: a missing or malformed file yields an empty cloud (no exception),
    before any device work."""
    c = spz.load_spz(str(tmp_path / "does_not_exist.spz"), spz.UnpackOptions())
    assert c.num_points == 0 and c.sh_degree == 0
    bad = tmp_path / "invalid.spz"
    bad.write_text("This is not a valid SPZ file")
    c = spz.load_spz(str(bad), spz.UnpackOptions())
    assert c.num_points == 0 and c.sh_degree == 0
    c = spz.load_splat_from_ply(str(tmp_path / "does_not_exist.ply"), spz.UnpackOptions())
    assert c.num_points == 0
    notply = tmp_path / "invalid.ply"
    notply.write_text("plyish\nformat ascii 1.0\nend_header\n")
    assert spz.load_splat_from_ply(str(notply), spz.UnpackOptions()).num_points == 0


def test_empty_cloud_saves_and_loads_without_a_device(tmp_path):
    """load_spz_test.py:753-772: a zero-point cloud is a 16-byte header inside gzip; no kernel runs,
    so this works on a machine with no GPU as well."""
    c = spz.GaussianCloud()
    fn = str(tmp_path / "empty_cloud.spz")
    assert spz.save_spz(c, spz.PackOptions(), fn) is True
    import gzip
    raw = gzip.decompress(open(fn, "rb").read())
    assert raw == bytes.fromhex("4e475350" "03000000" "00000000" "00" "0c" "00" "00")
    back = spz.load_spz(fn, spz.UnpackOptions())
    assert back.num_points == 0 and back.sh_degree == 0 and len(back.positions) == 0
    assert spz.save_spz(c, spz.PackOptions(), "/invalid/path/that/does/not/exist/test.spz") is False


def test_len_repr_and_sh_degree_range():
    """spz.cc:177-197: __len__ and num_points derive from positions, __repr__ text, sh_degree in [0, 3]."""
    c = spz.GaussianCloud()
    assert len(c) == 0 and repr(c) == "GaussianCloud(num_points=0, sh_degree=0, antialiased=False)"
    c.positions = np.zeros(6, np.float32)
    c.antialiased = True
    c.sh_degree = 3
    assert len(c) == 2 and c.num_points == 2
    assert repr(c) == "GaussianCloud(num_points=2, sh_degree=3, antialiased=True)"
    for bad in (-1, 4, 100):
        with pytest.raises(ValueError, match=r"sh_degree must be in \[0, 3\]"):
            c.sh_degree = bad
    assert c.sh_degree == 3
    with pytest.raises(TypeError):
        c.sh_degree = 1.5


def test_function_signatures_accept_the_shim_keywords(tmp_path):
    """spz.cc:347-361: keyword names and the defaulted `options` of the four module functions."""
    fn = str(tmp_path / "kw.spz")
    assert spz.save_spz(gaussians=spz.GaussianCloud(), options=spz.PackOptions(), filename=fn) is True
    assert spz.load_spz(filename=fn).num_points == 0                       # options defaults to UnpackOptions()
    assert spz.load_spz(fn, options=spz.UnpackOptions()).num_points == 0
    assert spz.load_splat_from_ply(filename=str(tmp_path / "missing.ply")).num_points == 0
    with pytest.raises(TypeError):
        spz.save_spz(spz.GaussianCloud(), fn)                               # options is required
    for name in ("load_spz", "save_spz", "load_splat_from_ply", "save_splat_to_ply"):
        assert getattr(spz, name).__doc__
