"""numpy restatement of RunningStandardScaler (TEST INFRASTRUCTURE: imported by tests/ only).

Reference: my_code/models/my_running_standard_scaler.py:57-119 -- fp64 running statistics (mean 0, variance 1, count 1
at the start), batch mean / torch.var (Bessel-corrected) over all leading dims, parallel-variance update, and
clamp((x - mean.float()) / (sqrt(var.float()) + eps), -clip, clip) in fp32 (or its inverse)."""
from __future__ import annotations

import numpy as np

F = np.float32


class RunningStandardScalerNP:
    def __init__(self, size: int, epsilon: float = 1e-8, clip_threshold: float = 5.0):
        self.epsilon, self.clip = epsilon, clip_threshold
        self.running_mean = np.zeros(size, np.float64)
        self.running_variance = np.ones(size, np.float64)
        self.current_count = 1.0

    def _parallel_variance(self, input_mean, input_var, input_count):       # :57-80
        delta = input_mean - self.running_mean
        total = self.current_count + input_count
        m2 = self.running_variance * self.current_count + input_var * input_count \
            + delta ** 2 * self.current_count * input_count / total
        self.running_mean = self.running_mean + delta * input_count / total
        self.running_variance = m2 / total
        self.current_count = total

    def compute(self, x, train=False, inverse=False):                      # :82-119
        x = np.asarray(x, np.float32)
        if train:
            flat = x.reshape(-1, x.shape[-1]).astype(np.float64)
            # the reference takes torch.mean / torch.var of the fp32 batch; fp64 accumulation here (the HIP path
            # accumulates in fp64 too): agreement with fp32 torch is a tolerance, stated in the tests
            self._parallel_variance(flat.mean(0), flat.var(0, ddof=1), flat.shape[0])
        mean32 = self.running_mean.astype(np.float32)
        std32 = np.sqrt(self.running_variance.astype(np.float32))
        if inverse:
            return (std32 * np.clip(x, F(-self.clip), F(self.clip)) + mean32).astype(np.float32)
        y = (x - mean32) / (std32 + F(self.epsilon))
        return np.clip(y, F(-self.clip), F(self.clip)).astype(np.float32)
