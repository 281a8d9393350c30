"""``RunningStandardScaler`` (``my_code/models/my_running_standard_scaler.py:19-140``): same constructor arguments,
buffers (``running_mean`` / ``running_variance`` / ``current_count``, fp64) and ``forward(x, train, inverse)``.

What differs is where the work runs: the batch moments are one HIP launch (``mdp_stats_moments``, fp64, merged across
ranks with one all-gather when a process group is up), and standardise + clip is ONE pass (``mdp_scaler_apply``) that can
write straight into a slot of the RL library's ``(T, N, O)`` rollout storage (``out=storage[t]``) -- the reference
spends five elementwise kernels and a temporary on it and the storage write is one more copy."""
from __future__ import annotations

import torch

from .. import _lib as L

try:
    import torch.distributed as dist
except Exception:  # pragma: no cover
    dist = None


class RunningStandardScaler:
    def __init__(self, size: int, epsilon: float = 1e-8, clip_threshold: float = 5.0, device=None, group=None):
        self.epsilon = float(epsilon)
        self.clip_threshold = float(clip_threshold)
        self.device = torch.device(device if device is not None else "cuda:0")
        self.size = int(size)
        self.group = group
        self.running_mean = torch.zeros(self.size, dtype=torch.float64, device=self.device)
        self.running_variance = torch.ones(self.size, dtype=torch.float64, device=self.device)
        self.current_count = torch.ones((), dtype=torch.float64, device=self.device)
        self.num_chunks = 32      # partial rows per batch: enough blocks to fill the chip at 4096 x 235
        self._partial = torch.zeros((self.num_chunks, 1 + 2 * self.size), dtype=torch.float64, device=self.device)
        self._mean32 = torch.zeros(self.size, dtype=torch.float32, device=self.device)
        self._std32 = torch.ones(self.size, dtype=torch.float32, device=self.device)
        self._denom = torch.ones(self.size, dtype=torch.float32, device=self.device)
        self._prepare()

    def _prepare(self):
        L.check(L.lib.mdp_scaler_prepare(self.size, self.running_mean.data_ptr(), self.running_variance.data_ptr(),
                                         self.epsilon, self._mean32.data_ptr(), self._std32.data_ptr(),
                                         self._denom.data_ptr(), L.stream(self.device)))

    @staticmethod
    def _rows(x: torch.Tensor, size: int, what: str):
        """(rows, ld) view of a (..., size) tensor whose leading dims collapse to one stride."""
        if x.dtype != torch.float32 or x.shape[-1] != size or x.stride(-1) != 1:
            raise ValueError(f"{what} must be float32 (..., {size}) with unit stride in the last dim, got {tuple(x.shape)}")
        if x.dim() == 1:
            return 1, size
        lead = x.reshape(-1, size) if x.is_contiguous() else None
        if lead is not None:
            return lead.shape[0], size
        if x.dim() == 2:
            return x.shape[0], x.stride(0)
        raise ValueError(f"{what}: leading dimensions must be contiguous")

    def _train(self, x: torch.Tensor):
        """my_running_standard_scaler.py:57-80, 94-105: batch mean / unbiased variance over all leading dims, then the
        parallel-variance update of the running statistics."""
        if x.dim() == 4:
            raise NotImplementedError("4-d inputs (one statistic for the whole tensor) are not on this path")
        rows, ld = self._rows(x, self.size, "x")
        L.check(L.lib.mdp_stats_moments_chunked(x.data_ptr(), rows, self.size, ld, self.num_chunks,
                                                self._partial.data_ptr(), L.stream(self.device)))
        parts = self._partial
        if dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            # one all-reduce(sum) of this rank's additive row about the shared running mean (distributed.shifted_sums):
            # the same collective shape as the episode statistics, which FusedStatistics packs behind it
            from ..distributed import shifted_sums

            row = shifted_sums(parts, self.running_mean)
            dist.all_reduce(row, op=dist.ReduceOp.SUM, group=self.group)
            self.apply_merged_row(row)
            return
        self._update_from(parts)

    def _update_from(self, parts: torch.Tensor):
        # merge + running update + fp32 casts: one launch, the statistics never leave the device
        L.check(L.lib.mdp_scaler_update(parts.shape[0], self.size, parts.data_ptr(), self.running_mean.data_ptr(),
                                        self.running_variance.data_ptr(), self.current_count.data_ptr(), self.epsilon,
                                        self._mean32.data_ptr(), self._std32.data_ptr(), self._denom.data_ptr(),
                                        L.stream(self.device)))

    def local_row(self, x: torch.Tensor) -> torch.Tensor:
        """This rank's batch as ONE additive fp64 row ``[n, sum (x - c), sum (x - c)^2]``, c = the running mean
        (``FusedStatistics`` packs it in front of the episode statistics)."""
        from ..distributed import shifted_sums

        rows, ld = self._rows(x, self.size, "x")
        L.check(L.lib.mdp_stats_moments_chunked(x.data_ptr(), rows, self.size, ld, self.num_chunks,
                                                self._partial.data_ptr(), L.stream(self.device)))
        return shifted_sums(self._partial, self.running_mean)

    def apply_merged_row(self, row: torch.Tensor):
        """Running update from the additive row summed over every rank."""
        from ..distributed import chan_row_from_shifted

        self._merged = chan_row_from_shifted(row, self.running_mean).contiguous()   # kept until the launch has read it
        self._update_from(self._merged)

    def forward(self, x: torch.Tensor, train: bool = False, inverse: bool = False, no_grad: bool = True,
                out: torch.Tensor | None = None) -> torch.Tensor:
        """my_running_standard_scaler.py:82-140.  ``out``: where the result goes (same shape as ``x``; e.g. one step of
        the rollout storage); a new tensor when omitted."""
        if train:
            self._train(x)
        rows, ld = self._rows(x, self.size, "x")
        if out is None:
            out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        elif out.shape != x.shape:
            raise ValueError(f"out shape {tuple(out.shape)} != x shape {tuple(x.shape)}")
        rows_o, ld_o = self._rows(out, self.size, "out")
        L.check(L.lib.mdp_scaler_apply(rows, self.size, x.data_ptr(), ld, self._mean32.data_ptr(), self._std32.data_ptr(),
                                       self._denom.data_ptr(), self.clip_threshold, 1 if inverse else 0, out.data_ptr(),
                                       ld_o, L.stream(self.device)))
        return out

    __call__ = forward
