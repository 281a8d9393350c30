"""Samplers of LAB/utils/math.py:1233-1421 on the device generator (``mdp_sample``).

* not gpu: the numpy restatement of each sampler's transformation against vectors the REFERENCE functions produced
  from recorded draws (tests/golden/samplers.npz);
* gpu: the HIP kernels fed the same recorded draws (``_inject``) against those vectors -- fp32 <= 1e-5 relative with an
  absolute floor of 2e-6 (a few ulp of values near 1: expf / logf / sinf / cosf differ from torch's CPU libm in the last
  bit) -- and the generator path itself through ranges, moments, seeding and epoch advance (the reference draws from
  torch's global generator, which cannot be matched number for number: distribution parity)."""
import numpy as np
import pytest

from conftest import assert_close, to_dev, to_np

TOL = dict(rtol=1e-5, atol=2e-6)


def test_oracle_samplers(golden):
    from oracle import samplers_np as O

    G = golden("samplers")
    u1, u2, z = G["u"][:, 0], G["u"][:, 1], G["z"]
    assert_close(O.sample_uniform(u1, -0.7, 2.5), G["uniform"], **TOL)
    assert_close(O.sample_uniform(u1, G["lo"], G["hi"]), G["uniform_tensor"], **TOL)
    assert_close(O.sample_log_uniform(u1, 1e-3, 10.0), G["log_uniform"], **TOL)
    assert_close(O.sample_triangle(u1, -1.0, 3.0), G["triangle"], **TOL)
    assert_close(O.sample_gaussian(z[:, 0], 0.3, 1.7), G["gaussian"], **TOL)
    assert_close(O.sample_gaussian(z[:, 0], G["lo"], G["hi"] - G["lo"]), G["gaussian_tensor"], **TOL)
    assert_close(O.sample_cylinder(u1, u2, 0.35, (-0.2, 0.9)), G["cylinder"], **TOL)
    assert_close(O.random_yaw_orientation(u1), G["yaw_quat"], **TOL)
    assert_close(O.random_orientation(z), G["quat"], **TOL)


@pytest.mark.gpu
def test_gpu_samplers_vs_reference_vectors(golden, device):
    from isaaclab_amd.utils import math as M

    G = golden("samplers")
    n = G["u"].shape[0]
    u = to_dev(G["u"], device)
    zpair = to_dev(np.stack([G["z"][:, 0], np.zeros(n, np.float32)], 1), device)
    lo, hi = to_dev(G["lo"], device), to_dev(G["hi"], device)
    dev = str(device)
    assert_close(to_np(M.sample_uniform(-0.7, 2.5, n, dev, _inject=u)), G["uniform"], **TOL)
    assert_close(to_np(M.sample_uniform(lo, hi, (n,), dev, _inject=u)), G["uniform_tensor"], **TOL)
    assert_close(to_np(M.sample_log_uniform(1e-3, 10.0, n, dev, _inject=u)), G["log_uniform"], **TOL)
    assert_close(to_np(M.sample_triangle(-1.0, 3.0, (n,), dev, _inject=u)), G["triangle"], **TOL)
    assert_close(to_np(M.sample_gaussian(0.3, 1.7, (n,), dev, _inject=zpair)), G["gaussian"], **TOL)
    assert_close(to_np(M.sample_gaussian(lo, hi - lo, (n,), dev, _inject=zpair)), G["gaussian_tensor"], **TOL)
    assert_close(to_np(M.sample_cylinder(0.35, (-0.2, 0.9), n, dev, _inject=u)), G["cylinder"], **TOL)
    assert_close(to_np(M.random_yaw_orientation(n, dev, _inject=u)), G["yaw_quat"], **TOL)
    assert_close(to_np(M.random_orientation(n, dev, _inject=to_dev(G["z"], device))), G["quat"], **TOL)
    q = to_np(M.default_orientation(5, dev))
    np.testing.assert_array_equal(q, np.tile(np.array([1, 0, 0, 0], np.float32), (5, 1)))


@pytest.mark.gpu
def test_gpu_samplers_generator(device):
    """Ranges and moments of the device generator (200 k samples: the standard error of a mean is ~2e-3 sigma), fresh
    numbers on every call, the same stream again after re-seeding, shapes."""
    import torch

    from isaaclab_amd.utils import math as M

    dev = str(device)
    n = 200_000
    M.seed_samplers(77, dev)
    x = to_np(M.sample_uniform(-2.0, 6.0, n, dev))
    assert x.min() >= -2.0 and x.max() < 6.0
    assert abs(x.mean() - 2.0) < 0.03 and abs(x.var() - 64.0 / 12.0) < 0.08
    hist = np.histogram(x, bins=16, range=(-2.0, 6.0))[0]
    assert np.abs(hist / n - 1 / 16).max() < 0.004
    y = to_np(M.sample_uniform(-2.0, 6.0, n, dev))
    assert not np.array_equal(x, y)                  # the epoch moved on the device
    M.seed_samplers(77, dev)
    np.testing.assert_array_equal(to_np(M.sample_uniform(-2.0, 6.0, n, dev)), x)     # same seed, same stream
    lg = np.log(to_np(M.sample_log_uniform(1e-3, 10.0, n, dev)))
    assert lg.min() >= np.log(1e-3) - 1e-5 and lg.max() <= np.log(10.0) + 1e-5
    assert abs(lg.mean() - 0.5 * (np.log(1e-3) + np.log(10.0))) < 0.03
    gs = to_np(M.sample_gaussian(0.5, 2.0, (n,), dev))
    assert abs(gs.mean() - 0.5) < 0.02 and abs(gs.std() - 2.0) < 0.02
    assert abs(((gs - 0.5) ** 3).mean()) < 0.2 and abs((((gs - 0.5) / 2.0) ** 4).mean() - 3.0) < 0.1
    mean = torch.linspace(-3, 3, 4096, device=device).repeat(40)
    gt = to_np(M.sample_gaussian(mean, torch.full_like(mean, 0.1), mean.shape, dev)) - to_np(mean)
    assert abs(gt.mean()) < 2e-3 and abs(gt.std() - 0.1) < 2e-3
    tr = to_np(M.sample_triangle(-1.0, 3.0, (400, 500), dev))
    assert tr.shape == (400, 500) and tr.min() >= -1.0 and tr.max() <= 3.0
    # the reference's map: r = sign(v) sqrt|v| of v ~ U(-1, 1), rescaled to [lower, upper]: mean at the centre,
    # variance (upper - lower)^2 / 8 (E r^2 = E|v| = 1/2, scale (upper - lower) / 2)
    assert abs(tr.mean() - 1.0) < 0.02 and abs(tr.var() - 16.0 / 8.0) < 0.03
    cy = to_np(M.sample_cylinder(0.35, (-0.2, 0.9), (100, 2000), dev))
    assert cy.shape == (100, 2000, 3)
    assert_close(np.hypot(cy[..., 0], cy[..., 1]), np.full((100, 2000), 0.35, np.float32), rtol=1e-5, atol=1e-6)
    assert cy[..., 2].min() >= -0.2 and cy[..., 2].max() <= 0.9 and abs(cy[..., 2].mean() - 0.35) < 0.01
    ang = np.arctan2(cy[..., 1], cy[..., 0])
    assert abs(ang.mean()) < 0.03 and abs(ang.var() - np.pi ** 2 / 3) < 0.05
    yq = to_np(M.random_yaw_orientation(n, dev))
    assert np.all(yq[:, 1:3] == 0.0)
    assert_close(np.linalg.norm(yq, axis=1), np.ones(n, np.float32), rtol=1e-5, atol=1e-6)
    yaw = 2.0 * np.arctan2(yq[:, 3], yq[:, 0])
    assert yaw.min() >= 0.0 and yaw.max() <= 2 * np.pi + 1e-5 and abs(yaw.mean() - np.pi) < 0.03
    rq = to_np(M.random_orientation(n, dev))
    assert_close(np.linalg.norm(rq, axis=1), np.ones(n, np.float32), rtol=1e-5, atol=1e-6)
    assert np.abs(rq.mean(0)).max() < 0.01 and np.abs((rq ** 2).mean(0) - 0.25).max() < 0.01
    with pytest.raises(ValueError, match="no CPU path"):
        M.sample_uniform(0.0, 1.0, 4, "cpu")
    # the draws follow torch's global seed like the reference's: manual_seed() after the first draw starts a new stream,
    # the same seed gives the same stream again; seed_samplers() pins a seed of its own
    import torch

    M._sampler_state.clear(); M._sampler_seed.clear()
    torch.manual_seed(123)
    a = to_np(M.sample_uniform(0.0, 1.0, 64, dev))
    torch.manual_seed(456)
    b = to_np(M.sample_uniform(0.0, 1.0, 64, dev))
    torch.manual_seed(123)
    c = to_np(M.sample_uniform(0.0, 1.0, 64, dev))
    assert np.array_equal(a, c) and not np.array_equal(a, b)
    # tensor bounds broadcast against `size` like `rand(*size) * (upper - lower) + lower`
    lo = torch.tensor([[0.0], [10.0], [20.0]], device=dev)
    x = to_np(M.sample_uniform(lo, lo + 1.0, (3, 5), dev))
    assert x.shape == (3, 5) and ((x >= to_np(lo)) & (x <= to_np(lo) + 1.0)).all()
    y = to_np(M.sample_uniform(torch.zeros(5, device=dev), torch.ones(5, device=dev), (3, 1), dev))
    assert y.shape == (3, 5)
    M.seed_samplers(1, dev)


@pytest.mark.gpu
def test_gpu_sampler_captured_in_a_graph_draws_fresh_numbers(device):
    import torch

    from isaaclab_amd.utils import math as M

    dev = str(device)
    M.seed_samplers(5, dev)
    M.sample_uniform(0.0, 1.0, 1024, dev)            # state tensor exists before capture
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = M.sample_uniform(0.0, 1.0, 1024, dev)
        g.replay()
        a = out.clone()
        g.replay()
        b = out.clone()
    torch.cuda.synchronize()
    assert not torch.equal(a, b)
