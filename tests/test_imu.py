"""Imu against golden vectors from the reference class (tests/golden/make_golden.py::gen_imu: three
updates, the middle one for a subset of envs).  CPU: the oracle.  GPU: the Imu mirror / mdp_imu_update, 1e-5 relative
(accelerations divide velocity differences by dt = 5 ms: 1e-3 absolute floor on those)."""
import os

import numpy as np
import pytest
import torch

from conftest import assert_close, to_np
from oracle.imu_np import ImuNP

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "imu.npz"))
NAMES = ("pos_w", "quat_w", "lin_vel_b", "ang_vel_b", "lin_acc_b", "ang_acc_b")


def _ids(k, n):
    return G["subset"].astype(np.int64) if k == 1 else np.arange(n)


def test_oracle_imu_against_reference_vectors():
    n = G["coms"].shape[0]
    orc = ImuNP(n, G["offset_pos"], G["offset_quat"])
    for k in range(3):
        orc.update(_ids(k, n), G["transforms"][k], G["velocities"][k], G["coms"], 0.005)
        for name in NAMES:
            tol = dict(rtol=1e-5, atol=2e-3) if "acc" in name else dict(rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(getattr(orc, name), G[f"{name}_{k}"], err_msg=f"{name} update {k}", **tol)
        np.testing.assert_allclose(orc.prev_lin, G[f"prev_lin_{k}"], rtol=1e-6, atol=1e-6)
        np.testing.assert_array_equal(orc.prev_ang, G[f"prev_ang_{k}"])


@pytest.mark.gpu
def test_gpu_imu_against_reference_vectors(device):
    from isaaclab_amd.sensors import Imu, ImuCfg

    n = G["coms"].shape[0]
    state = {}
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)

    class View:
        count = n
        get_transforms = lambda self: state["t"]
        get_velocities = lambda self: state["v"]
        get_coms = lambda self: state["c"]

    state["c"] = dev(G["coms"])
    cfg = ImuCfg(update_period=0.0, offset=ImuCfg.OffsetCfg(pos=tuple(G["offset_pos"][0].tolist()), rot=tuple(G["offset_quat"][0].tolist())))
    imu = Imu(cfg, View(), n, str(device), 0.005)
    with pytest.raises(RuntimeError, match="update function must be called"):
        imu.data
    for k in range(3):
        state["t"], state["v"] = dev(G["transforms"][k]), dev(G["velocities"][k])
        imu.update(0.005)
        if k == 1:                                   # only the subset is evaluated this time
            keep = torch.ones(n, dtype=torch.bool, device=device)
            keep[dev(G["subset"]).long()] = False
            imu._is_outdated[keep] = False
        d = imu.data
        for name in NAMES:
            tol = dict(rtol=1e-5, atol=2e-3) if "acc" in name else dict(rtol=1e-5, atol=1e-6)
            assert_close(to_np(getattr(d, name)), G[f"{name}_{k}"], msg=f"{name} update {k}", **tol)
        assert not imu._is_outdated.any()
        assert_close(to_np(imu._prev_lin_vel_w), G[f"prev_lin_{k}"], rtol=1e-6, atol=1e-6)
    imu.reset([1, 5])
    assert (imu._data.lin_acc_b[[1, 5]] == 0).all() and imu._is_outdated[[1, 5]].all() and (imu._timestamp[[1, 5]] == 0).all()
