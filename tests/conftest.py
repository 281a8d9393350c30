"""Shared test plumbing.

* ``-m "not gpu"``: oracle vs the committed reference-generated fixtures (tests/golden/*.npz), host logic,
  ABI export check.  Runs on CPU in the build container.
* ``-m gpu``: parity tests proper; they call the HIP kernels through the C ABI (ctypes) and compare with the
  oracle and the same fixtures.  Nothing here reads /root/reference.

Tolerances (BASELINE.json north_star): masks / ids / face ids bit-exact, fp32 values <= 1e-5 relative.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
RTOL = 1.0e-5          # the north-star fp32 tolerance
ATOL = 2.0e-6          # absolute floor for values that cancel to ~0 (sums of O(1) terms)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu on the GPU box")


def load_golden(name: str) -> dict:
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get


def assert_close(actual, expected, rtol=RTOL, atol=ATOL, msg=""):
    a = np.asarray(actual)
    e = np.asarray(expected)
    assert a.shape == e.shape, f"{msg}: shape {a.shape} vs {e.shape}"
    if e.dtype == np.bool_ or np.issubdtype(e.dtype, np.integer):
        np.testing.assert_array_equal(a, e, err_msg=msg)
    else:
        np.testing.assert_allclose(a.astype(np.float64), e.astype(np.float64), rtol=rtol, atol=atol, err_msg=msg,
                                   equal_nan=True)


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


def to_dev(x, device):
    import torch

    return torch.from_numpy(np.ascontiguousarray(x)).to(device)


def to_np(t):
    return t.detach().cpu().numpy()
