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


def _record(a, e, rtol, atol, msg):
    """MDP_PARITY_TABLE=<file>: one line per floating-point comparison -- worst absolute error, the value it occurred
    at, its size in units in the last place of that value, worst relative error and the absolute floor the comparison
    would need at rtol = 1e-5 (<= 0: none).  profiles/r02_parity_error_table.txt is such a run on the MI355X."""
    path = os.environ.get("MDP_PARITY_TABLE")
    if not path or a.size == 0:
        return
    a64, e64 = a.astype(np.float64), e.astype(np.float64)
    fin = np.isfinite(a64) & np.isfinite(e64)
    if not fin.any():
        return
    d = np.where(fin, np.abs(a64 - e64), 0.0)
    i = np.unravel_index(d.argmax(), d.shape)
    ulp = np.spacing(np.abs(e[i]).astype(np.float32)) if e.dtype == np.float32 else np.spacing(np.float32(abs(e64[i])))
    rel = np.where(fin & (e64 != 0), d / np.maximum(np.abs(e64), 1e-300), 0.0)
    need = float(np.max(np.where(fin, d - RTOL * np.abs(e64), -np.inf)))
    test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
    with open(path, "a") as f:
        f.write(f"{test}\t{msg}\tmax_abs={d.max():.3e}\tat={e64[i]:.5g}\tulps={d.max() / float(ulp):.1f}"
                f"\tmax_rel={rel.max():.3e}\tatol_needed_at_rtol_1e-5={need:.3e}\tused_rtol={rtol:g}\tused_atol={atol:g}\n")


def assert_close(actual, expected, rtol=RTOL, atol=ATOL, msg=""):
    a = np.asarray(actual)
    e = np.asarray(expected)
    assert a.shape == e.shape, f"{msg}: shape {a.shape} vs {e.shape}"
    if not (e.dtype == np.bool_ or np.issubdtype(e.dtype, np.integer)):
        _record(a, e, rtol, atol, msg)
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


# Observation floors (the contract is <= 1e-5 RELATIVE; a floor is only needed where a term cancels towards zero):
#  * every column but the height scan: 4e-6 absolute -- two ulps of the O(10) operands (joint positions, velocities,
#    commands) a column is the difference or rotation of;
#  * height-scan columns (the last 187): (sensor z - hit z) - 0.5 with the hit z taken from a ray that starts 20 m above the
#    sensor, t = 20 +- 1 and its last bit is 1.9e-6: the measured worst case is four of them, 7.6e-6
#    (profiles/r02_parity_error_table.txt, tools/soak_env_step.py), i.e. 4e-7 relative to the operand the error comes
#    from; the floor is 1e-5 absolute = 5e-7 of that operand.
OBS_ATOL = 4.0e-6
SCAN_ATOL = 1.0e-5


def assert_obs_close(got, exp, scan_cols: int = 0, msg: str = ""):
    """Policy observation (N, O) against the oracle's: rtol 1e-5 everywhere, per-column-family absolute floors."""
    got, exp = np.asarray(got), np.asarray(exp)
    assert got.shape == exp.shape, f"{msg}: shape {got.shape} vs {exp.shape}"
    split = got.shape[1] - scan_cols
    assert_close(got[:, :split], exp[:, :split], rtol=RTOL, atol=OBS_ATOL, msg=msg + " (non-scan columns)")
    if scan_cols:
        assert_close(got[:, split:], exp[:, split:], rtol=RTOL, atol=SCAN_ATOL, msg=msg + " (height-scan columns)")
