import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
FIELDS = ("positions", "scales", "rotations", "alphas", "colors", "sh")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def bits(a):
    """float32 array -> uint32 bit patterns (bit-exact comparison, NaN payloads and -0 included)."""
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bits_equal(got, want, what=""):
    g, w = bits(got), bits(want)
    assert g.shape == w.shape, f"{what}: shape {g.shape} != {w.shape}"
    if not np.array_equal(g, w):
        idx = np.nonzero(g != w)[0]
        i = int(idx[0])
        raise AssertionError(f"{what}: {idx.size} of {g.size} floats differ; first at {i}: "
                             f"got {np.float32(got.reshape(-1)[i])!r} (0x{g[i]:08x}) "
                             f"want {np.float32(np.asarray(want).reshape(-1)[i])!r} (0x{w[i]:08x})")


def assert_bytes_equal(got, want, what=""):
    g = np.ascontiguousarray(got, dtype=np.uint8).reshape(-1)
    w = np.ascontiguousarray(want, dtype=np.uint8).reshape(-1)
    assert g.shape == w.shape, f"{what}: size {g.size} != {w.size}"
    if not np.array_equal(g, w):
        idx = np.nonzero(g != w)[0]
        i = int(idx[0])
        raise AssertionError(f"{what}: {idx.size} of {g.size} bytes differ; first at offset {i}: "
                             f"got 0x{g[i]:02x} want 0x{w[i]:02x}")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle.pyoracle import Reference, REF_SO
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libspz_ref.so not built (needs /root/reference at build time)")
    return Reference()


@pytest.fixture(scope="session")
def cuda():
    # one non-polling check: a missing device fails at once, with the HIP error code
    import torch
    from spz_amd import abi
    if abi.load_library().spz_amd_device_count() < 1 or not torch.cuda.is_available():
        pytest.fail("test marked gpu but the HIP runtime reports no device "
                    f"(hipError {abi.load_library().spz_amd_last_hip_error()})")
    return torch.device("cuda:0")
