"""RunningStandardScaler.  CPU: the oracle restatement against the reference's expressions evaluated
with torch (my_running_standard_scaler.py:57-119).  GPU: the HIP path (mdp_stats_moments + mdp_scaler_prepare +
mdp_scaler_apply) against the oracle: standardised values within 1e-5 relative (1e-6 absolute), running statistics
within 1e-9 relative (fp64 on both sides); the clip bounds are hit exactly."""
import numpy as np
import pytest
import torch

from conftest import assert_close, to_np
from oracle.scaler_np import RunningStandardScalerNP


def _batches(seed=0, steps=5, n=257, d=37):
    rng = np.random.RandomState(seed)
    scale = rng.uniform(0.01, 30.0, size=d).astype(np.float32)
    shift = rng.normal(0, 5.0, size=d).astype(np.float32)
    xs = [(rng.normal(size=(n, d)).astype(np.float32) * scale + shift) for _ in range(steps)]
    if steps > 2:
        xs[2][3, :5] = 1e6      # outliers: the clip must bite
    return xs


class _TorchScaler:
    """The reference's statements, verbatim in structure, on CPU torch."""

    def __init__(self, size, epsilon=1e-8, clip_threshold=5.0):
        self.epsilon, self.clip_threshold = epsilon, clip_threshold
        self.running_mean = torch.zeros(size, dtype=torch.float64)
        self.running_variance = torch.ones(size, dtype=torch.float64)
        self.current_count = torch.ones((), dtype=torch.float64)

    def _parallel_variance(self, input_mean, input_var, input_count):
        delta = input_mean - self.running_mean
        total_count = self.current_count + input_count
        M2 = (self.running_variance * self.current_count) + (input_var * input_count) \
            + delta**2 * self.current_count * input_count / total_count
        self.running_mean = self.running_mean + delta * input_count / total_count
        self.running_variance = M2 / total_count
        self.current_count = total_count

    def compute(self, x, train=False, inverse=False):
        if train:
            if x.dim() == 3:
                self._parallel_variance(torch.mean(x, dim=(0, 1)), torch.var(x, dim=(0, 1)), x.shape[0] * x.shape[1])
            else:
                self._parallel_variance(torch.mean(x, dim=0), torch.var(x, dim=0), x.shape[0])
        if inverse:
            return torch.sqrt(self.running_variance.float()) \
                * torch.clamp(x, min=-self.clip_threshold, max=self.clip_threshold) + self.running_mean.float()
        return torch.clamp((x - self.running_mean.float()) / (torch.sqrt(self.running_variance.float()) + self.epsilon),
                           min=-self.clip_threshold, max=self.clip_threshold)


def test_oracle_matches_reference_expressions():
    xs = _batches()
    orc, ref = RunningStandardScalerNP(37), _TorchScaler(37)
    for t, x in enumerate(xs):
        y, yr = orc.compute(x, train=True), ref.compute(torch.from_numpy(x), train=True).numpy()
        # the reference reduces the fp32 batch in fp32: statistics agree to fp32 reduction error, outputs follow
        np.testing.assert_allclose(orc.running_mean, ref.running_mean.numpy(), rtol=2e-5, atol=1e-5)
        np.testing.assert_allclose(orc.running_variance, ref.running_variance.numpy(), rtol=5e-5)
        np.testing.assert_allclose(y, yr, rtol=1e-4, atol=2e-4)
        assert y.max() == 5.0 or t != 2
    x3 = np.stack(xs[:3])
    orc.compute(x3, train=True)
    ref.compute(torch.from_numpy(x3), train=True)
    np.testing.assert_allclose(orc.running_variance, ref.running_variance.numpy(), rtol=5e-5)
    back, back_r = orc.compute(y, inverse=True), ref.compute(torch.from_numpy(y), inverse=True).numpy()
    np.testing.assert_allclose(back, back_r, rtol=1e-4, atol=1e-4)
    # with identical statistics the elementwise map is the same fp32 expression: exact
    orc.running_mean, orc.running_variance = ref.running_mean.numpy().copy(), ref.running_variance.numpy().copy()
    np.testing.assert_array_equal(orc.compute(xs[0]), ref.compute(torch.from_numpy(xs[0])).numpy())
    np.testing.assert_array_equal(orc.compute(y, inverse=True), ref.compute(torch.from_numpy(y), inverse=True).numpy())


@pytest.mark.gpu
def test_gpu_scaler_matches_oracle(device):
    from isaaclab_amd.rl import RunningStandardScaler

    xs = _batches(seed=3, steps=6, n=513, d=235)
    orc = RunningStandardScalerNP(235)
    sc = RunningStandardScaler(235, device=device)
    storage = torch.zeros((len(xs), 513, 235 + 13), device=device)      # rollout storage with wider rows
    for t, x in enumerate(xs):
        xd = torch.from_numpy(x).to(device)
        exp = orc.compute(x, train=True)
        got = sc(xd, train=True, out=storage[t, :, :235])
        assert got.data_ptr() == storage[t].data_ptr()
        assert_close(to_np(got), exp, rtol=1e-5, atol=1e-6, msg=f"step {t}")
        np.testing.assert_allclose(to_np(sc.running_mean), orc.running_mean, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(to_np(sc.running_variance), orc.running_variance, rtol=1e-9)
        assert float(sc.current_count) == orc.current_count
    assert (storage[..., 235:] == 0).all() and float(storage[2].max()) == 5.0
    # evaluation mode, 3-d input, inverse
    x3 = np.stack(xs[:2])
    assert_close(to_np(sc(torch.from_numpy(x3).to(device))), orc.compute(x3), rtol=1e-5, atol=1e-6)
    y = orc.compute(xs[1])
    assert_close(to_np(sc(torch.from_numpy(y).to(device), inverse=True)), orc.compute(y, inverse=True), rtol=1e-5, atol=1e-5)
    exp3 = orc.compute(x3, train=True)
    assert_close(to_np(sc(torch.from_numpy(x3).to(device), train=True)), exp3, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(to_np(sc.running_variance), orc.running_variance, rtol=1e-9)
    with pytest.raises(ValueError):
        sc(torch.zeros((4, 7), device=device))


@pytest.mark.gpu
def test_gpu_scaler_merges_partials_of_several_ranks(device):
    """The cross-rank form of the update: every rank's chunk partials (mdp_stats_moments_chunked on its env shard) are
    gathered and merged by ONE mdp_scaler_update -- here two shards of unequal size on one GPU -- and must give the
    statistics of the whole batch (the oracle on the concatenation)."""
    from isaaclab_amd import _lib as L
    from isaaclab_amd.rl import RunningStandardScaler

    x = _batches(seed=5, steps=1, n=1001, d=48)[0]
    orc = RunningStandardScalerNP(48)
    exp = orc.compute(x, train=True)
    sc = RunningStandardScaler(48, device=device)
    xd = torch.from_numpy(x).to(device)
    parts = []
    for lo, hi in ((0, 387), (387, 1001)):
        p = torch.zeros((sc.num_chunks, 1 + 2 * 48), dtype=torch.float64, device=device)
        shard = xd[lo:hi]
        L.check(L.lib.mdp_stats_moments_chunked(shard.data_ptr(), hi - lo, 48, 48, sc.num_chunks, p.data_ptr(), L.stream(device)))
        parts.append(p)
    allp = torch.cat(parts)
    assert float(allp[:, 0].sum()) == 1001
    L.check(L.lib.mdp_scaler_update(allp.shape[0], 48, allp.data_ptr(), sc.running_mean.data_ptr(),
                                    sc.running_variance.data_ptr(), sc.current_count.data_ptr(), sc.epsilon,
                                    sc._mean32.data_ptr(), sc._std32.data_ptr(), sc._denom.data_ptr(), L.stream(device)))
    np.testing.assert_allclose(to_np(sc.running_mean), orc.running_mean, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(to_np(sc.running_variance), orc.running_variance, rtol=1e-9)
    assert float(sc.current_count) == 1002.0
    assert_close(to_np(sc(xd)), exp, rtol=1e-5, atol=1e-6)
