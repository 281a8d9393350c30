"""GPU: ``isaaclab_amd.utils.math`` (HIP, through ``mdp_math_op`` of the C ABI) against the reference-generated
fixtures and the numpy oracle.  Tolerance: fp32 <= 1e-5 relative with the 2e-6 absolute floor of conftest.py (8 ulp of
pi, for angles that cancel to ~0; measured worst cases in profiles/r02_parity_error_table.txt), sign selections
bit-exact via the values themselves."""
import numpy as np
import pytest

from conftest import assert_close, to_np
from test_oracle_math import SIZES, math_cases

pytestmark = pytest.mark.gpu


class DeviceMath:
    """Calls isaaclab_amd.utils.math with device tensors, returns numpy."""

    def __init__(self, device):
        import torch

        import isaaclab_amd.utils.math as m

        self.m, self.torch, self.device = m, torch, device

    def __getattr__(self, name):
        fn = getattr(self.m, name)
        torch = self.torch

        def call(*args):
            conv = [torch.from_numpy(np.ascontiguousarray(a)).to(self.device) if isinstance(a, np.ndarray) else a
                    for a in args]
            out = fn(*conv)
            if isinstance(out, tuple):
                return tuple(to_np(o) for o in out)
            return to_np(out)

        return call


@pytest.mark.parametrize("n", SIZES)
def test_math_vs_reference_fixtures(golden, device, n):
    G = golden("math")
    dm = DeviceMath(device)
    for name, fn in math_cases(G, n):
        assert_close(fn(dm), G[f"n{n}_{name}"], rtol=1e-5, atol=2e-6, msg=f"{name} n={n}")


def test_transform_points(golden, device):
    G = golden("math")
    dm = DeviceMath(device)
    for n in SIZES:
        p = f"n{n}_"
        assert_close(dm.transform_points(G[p + "pts"], G[p + "t1"], G[p + "q1"]), G[p + "transform_points"],
                     rtol=1e-5, atol=2e-6, msg=f"transform_points n={n}")


def test_math_vs_oracle_large(device):
    """4096 x 17 bodies worth of quaternions: HIP vs the numpy oracle on the same seeded inputs."""
    from oracle import math_np as M

    rng = np.random.default_rng(5)
    n = 4096 * 17
    q = rng.standard_normal((n, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q2 = np.roll(q, 1, axis=0)
    v = rng.standard_normal((n, 3)).astype(np.float32)
    dm = DeviceMath(device)
    assert_close(dm.quat_rotate_inverse(q, v), M.quat_rotate_inverse(q, v), rtol=1e-5, atol=2e-6)
    assert_close(dm.quat_mul(q, q2), M.quat_mul(q, q2), rtol=1e-5, atol=2e-6)
    assert_close(dm.yaw_quat(q), M.yaw_quat(q), rtol=1e-5, atol=2e-6)
    assert_close(dm.quat_apply_yaw(q, v), M.quat_apply_yaw(q, v), rtol=1e-5, atol=4e-6)
    ang = (rng.random((n,)).astype(np.float32) - 0.5) * 100
    assert_close(dm.wrap_to_pi(ang), M.wrap_to_pi(ang), rtol=1e-5, atol=4e-6, msg="wrap_to_pi |x| <= 50 (ulp of 50 = 3.8e-6)")
    # size-independent properties: rotate o rotate_inverse = id, |q1*q2| = 1
    back = dm.quat_rotate(q, dm.quat_rotate_inverse(q, v))
    assert_close(back, v, rtol=1e-5, atol=4e-6)
    assert_close(np.linalg.norm(dm.quat_mul(q, q2), axis=1), np.ones(n, np.float32), atol=1e-6)


def test_empty_and_errors(device):
    import torch

    import isaaclab_amd.utils.math as m

    out = m.quat_mul(torch.zeros(0, 4, device=device), torch.zeros(0, 4, device=device))
    assert out.shape == (0, 4)
    with pytest.raises(ValueError):
        m.quat_mul(torch.zeros(3, 4, device=device), torch.zeros(4, 4, device=device))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.quat_mul(torch.zeros(3, 4), torch.zeros(3, 4))
