"""SURVEY §8f row 1 — .ply rows <-> GaussianCloud arrays.

CPU part: the oracle's restatement of the reference's PLY loops against files the reference itself
wrote and what the reference loaded from them (tests/golden/ply.npz).
GPU part (-m gpu): the HIP shuffles (spz_ply_kernels.hip, through the C ABI) against the oracle
bit for bit, and the Python module's files against the reference's files byte for byte."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import FIELDS, assert_bits_equal, assert_bytes_equal, load_golden

NAMES = ("two_sh3", "two_sh0", "n131_sh1", "n131_sh2", "n131_sh3")
SH_DIM = {0: 0, 1: 3, 2: 8, 3: 15}


def split_ply(blob):
    """-> (property names, float32 rows flat, vertex count) of a binary-LE .ply held in a uint8 array."""
    data = blob.tobytes()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    props, n = [], None
    for line in data[:end].decode().splitlines():
        line = line.strip()
        if line.startswith("property float "):
            props.append(line[len("property float "):])
        elif line.startswith("element vertex "):
            n = int(line[len("element vertex "):])
    return props, np.frombuffer(data[end:], np.float32), n


def columns_from(props):
    from spz_amd import abi
    c = abi.PlyColumns()
    c.stride = len(props)
    ix = {p: i for i, p in enumerate(props)}
    for k, nm in enumerate(("x", "y", "z")):
        c.position[k] = ix[nm]
    for k in range(3):
        c.scale[k] = ix[f"scale_{k}"]
        c.color[k] = ix[f"f_dc_{k}"]
    for k, nm in enumerate(("rot_1", "rot_2", "rot_3", "rot_0")):
        c.rotation[k] = ix[nm]
    c.alpha = ix["opacity"]
    rest = []
    while f"f_rest_{len(rest)}" in ix and len(rest) < 45:
        rest.append(ix[f"f_rest_{len(rest)}"])
    c.sh_dim = len(rest) // 3
    for k in range(c.sh_dim * 3):
        c.sh[k] = rest[k]
    return c


def deg_of(name):
    return int(name[-1])


@pytest.mark.parametrize("name", NAMES)
def test_oracle_rows_equal_reference_files(oracle, name):
    g = load_golden("ply.npz")
    c = {k: g[f"{name}_in_{k}"] for k in FIELDS}
    n = c["alphas"].size
    shd = SH_DIM[deg_of(name)]
    for frm in (0, 4, 7):
        props, rows, nv = split_ply(g[f"{name}_file_from{frm}"])
        assert nv == n and len(props) == 17 + 3 * shd
        assert_bits_equal(oracle.cloud_to_ply_rows(c, n, shd, frm), rows, f"{name} from={frm}")
    props, rows, _ = split_ply(g[f"{name}_file_from0"])
    for to in (0, 4, 7):
        u = oracle.ply_rows_to_cloud(rows, n, columns_from(props), to)
        for k in FIELDS:
            assert_bits_equal(u[k], g[f"{name}_load_to{to}_{k}"], f"{name} to={to} {k}")


def test_oracle_hand_made_file(oracle):
    g = load_golden("ply.npz")
    props, rows, n = split_ply(g["odd_file"])
    cols = columns_from(props)
    assert (cols.stride, cols.sh_dim, n) == (23, 2, 70)
    for to in (0, 4, 7):
        info = g[f"odd_load_to{to}_info"]
        assert info.tolist() == [70, 0, 70 * 6]  # degreeForDim(2) = 0, yet 6 sh floats per point are kept
        u = oracle.ply_rows_to_cloud(rows, n, cols, to)
        for k in FIELDS:
            assert_bits_equal(u[k], g[f"odd_load_to{to}_{k}"], f"odd to={to} {k}")


def test_default_columns_match_the_writer_layout():
    from spz_amd import abi
    L = abi.load_library()
    g = load_golden("ply.npz")
    for name in NAMES:
        props, _, _ = split_ply(g[f"{name}_file_from0"])
        want = columns_from(props)
        got = abi.PlyColumns()
        assert L.spz_amd_ply_default_columns(SH_DIM[deg_of(name)], C.byref(got)) == 0
        assert bytes(got) == bytes(want), name
    bad = abi.PlyColumns()
    assert L.spz_amd_ply_default_columns(16, C.byref(bad)) == abi.ERR_INVALID_ARG


# ------------------------------------------------------------------------------------------------
# GPU
# ------------------------------------------------------------------------------------------------
def _ptrs(d):
    from spz_amd import abi
    return abi.CloudPtrs(*[d[k].data_ptr() if d[k].numel() else None for k in FIELDS])


@pytest.mark.gpu
@pytest.mark.parametrize("n,shd", [(1, 0), (63, 3), (64, 8), (65, 15), (131, 15), (100_003, 15), (70_001, 5), (3, 1)])
def test_gpu_row_shuffles_match_oracle(cuda, oracle, n, shd):
    import torch
    from spz_amd import abi
    L = abi.load_library()
    rng = np.random.default_rng(n + shd)
    c = dict(positions=rng.standard_normal(n * 3), scales=rng.standard_normal(n * 3),
             rotations=rng.standard_normal(n * 4), alphas=rng.standard_normal(n), colors=rng.standard_normal(n * 3),
             sh=rng.standard_normal(n * shd * 3))
    c = {k: v.astype(np.float32) for k, v in c.items()}
    c["positions"][:2] = [0.0, -0.0]
    t = {k: torch.from_numpy(c[k]).to(cuda) for k in FIELDS}
    D = 17 + 3 * shd
    rows = torch.full((n * D,), 7.0, dtype=torch.float32, device=cuda)
    for frm in (0, 4, 7, 1):
        abi.check(L.spz_amd_cloud_to_ply_rows_device(C.byref(_ptrs(t)), n, shd, frm, rows.data_ptr(), None), "to_rows")
        torch.cuda.synchronize()
        assert_bits_equal(rows.cpu().numpy(), oracle.cloud_to_ply_rows(c, n, shd, frm), f"rows from={frm}")
    # rows -> cloud, default layout and a shuffled layout with extra columns
    cols = abi.PlyColumns()
    L.spz_amd_ply_default_columns(shd, C.byref(cols))
    perm = rng.permutation(D + 3)
    wide = np.zeros((n, D + 3), np.float32)
    base = rows.cpu().numpy().reshape(n, D)
    wide[:, perm[:D]] = base
    wide[:, perm[D:]] = rng.standard_normal((n, 3)).astype(np.float32)
    cols2 = abi.PlyColumns()
    cols2.stride, cols2.sh_dim = D + 3, shd
    for k in range(3):
        cols2.position[k], cols2.scale[k], cols2.color[k] = perm[cols.position[k]], perm[cols.scale[k]], perm[cols.color[k]]
    for k in range(4):
        cols2.rotation[k] = perm[cols.rotation[k]]
    cols2.alpha = perm[cols.alpha]
    for k in range(3 * shd):
        cols2.sh[k] = perm[cols.sh[k]]
    for cc, r in ((cols, base.reshape(-1)), (cols2, wide.reshape(-1))):
        d_rows = torch.from_numpy(np.ascontiguousarray(r)).to(cuda)
        out = {k: torch.full_like(t[k], 5.0) for k in FIELDS}
        for to in (0, 4, 7, 2):
            abi.check(L.spz_amd_ply_rows_to_cloud_device(d_rows.data_ptr(), n, C.byref(cc), to, C.byref(_ptrs(out)), None),
                      "to_cloud")
            torch.cuda.synchronize()
            want = oracle.ply_rows_to_cloud(r, n, cc, to)
            for k in FIELDS:
                assert_bits_equal(out[k].cpu().numpy(), want[k], f"stride={cc.stride} to={to} {k}")


@pytest.mark.gpu
def test_gpu_python_module_files_equal_reference_files(cuda, tmp_path):
    import spz_amd.spz as spz
    g = load_golden("ply.npz")
    for name in NAMES:
        c = spz.GaussianCloud()
        c.sh_degree = deg_of(name)
        for k in FIELDS:
            setattr(c, k, g[f"{name}_in_{k}"])
        for frm in (0, 4, 7):
            o = spz.PackOptions()
            o.from_coord = spz.CoordinateSystem(frm)
            f = str(tmp_path / f"{name}_{frm}.ply")
            assert spz.save_splat_to_ply(c, o, f) is True
            assert_bytes_equal(np.frombuffer(open(f, "rb").read(), np.uint8), g[f"{name}_file_from{frm}"], f"{name} {frm}")
        f = str(tmp_path / f"{name}_ref.ply")
        open(f, "wb").write(g[f"{name}_file_from0"].tobytes())
        for to in (0, 4, 7):
            u = spz.UnpackOptions()
            u.to_coord = spz.CoordinateSystem(to)
            d = spz.load_splat_from_ply(f, u)
            assert d.num_points == c.num_points and d.sh_degree == deg_of(name)
            for k in FIELDS:
                assert_bits_equal(getattr(d, k), g[f"{name}_load_to{to}_{k}"], f"{name} to={to} {k}")
    f = str(tmp_path / "odd.ply")
    open(f, "wb").write(g["odd_file"].tobytes())
    for to in (0, 4, 7):
        u = spz.UnpackOptions()
        u.to_coord = spz.CoordinateSystem(to)
        d = spz.load_splat_from_ply(f, u)
        assert [d.num_points, d.sh_degree, len(d.sh)] == g[f"odd_load_to{to}_info"].tolist()
        for k in FIELDS:
            assert_bits_equal(getattr(d, k), g[f"odd_load_to{to}_{k}"], f"odd to={to} {k}")
    # error conventions of loadSplatFromPly (load-spz.cc:693-726): empty cloud, never an exception
    bad = str(tmp_path / "bad.ply")
    for blob in (b"plx\n", b"ply\nformat ascii 1.0\n", b"ply\nformat binary_little_endian 1.0\nelement vertex 0\nend_header\n",
                 b"ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty double x\nend_header\n",
                 b"ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty float x\nend_header\n"):
        open(bad, "wb").write(blob)
        assert spz.load_splat_from_ply(bad, spz.UnpackOptions()).num_points == 0
    assert spz.load_splat_from_ply(str(tmp_path / "missing.ply"), spz.UnpackOptions()).num_points == 0


@pytest.mark.gpu
def test_gpu_large_ply_files_round_trip(cuda, oracle, tmp_path):
    """A 400 k-point SH3 cloud as a .ply (99 MB: above the size from which loadSplatFromPly reads the vertex rows by
    several threads into an unwritten buffer, and saveSplatToPly no longer zero-fills its rows): the file is header + the
    oracle's rows (load-spz.cc:856-893), loading it gives the oracle's arrays for the rows (:814-842) for two target
    systems, and a file cut short is the reference's error (an empty cloud), not a crash."""
    import spz_amd.spz as spz
    from spz_amd.synth import make_cloud_numpy
    n, deg, shd = 400_000, 3, 15
    c = make_cloud_numpy(n, deg, 5)
    g = spz.GaussianCloud()
    g.sh_degree = deg
    for k in FIELDS:
        setattr(g, k, c[k])
    o = spz.PackOptions()
    o.from_coord = spz.RUB
    f = str(tmp_path / "big.ply")
    assert spz.save_splat_to_ply(g, o, f) is True
    blob = open(f, "rb").read()
    props, rows, count = split_ply(np.frombuffer(blob, np.uint8))
    assert count == n and rows.size == n * (17 + 3 * shd)
    want_rows = oracle.cloud_to_ply_rows(c, n, shd, int(spz.RUB))
    assert_bits_equal(rows, want_rows, "rows")
    cols = columns_from(props)
    for to in (int(spz.RDF), int(spz.LUF)):
        u = spz.UnpackOptions()
        u.to_coord = spz.CoordinateSystem(to)
        d = spz.load_splat_from_ply(f, u)
        assert d.num_points == n and d.sh_degree == deg
        want = oracle.ply_rows_to_cloud(want_rows, n, cols, to)
        for k in FIELDS:
            assert_bits_equal(getattr(d, k), want[k], f"to={to} {k}")
    open(f, "wb").write(blob[:len(blob) - 4096])
    assert spz.load_splat_from_ply(f, spz.UnpackOptions()).num_points == 0
