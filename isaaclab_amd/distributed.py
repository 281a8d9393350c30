"""Multi-GPU layer of the MDP path: one process per GPU, envs sharded by contiguous
ranges, NO data-path collective.  The only cross-rank traffic is statistics:

* episode statistics (``Episode_Reward/*``, ``Episode_Termination/*``, ``Metrics/base_velocity/*``;
  reward_manager.py:115-121, termination_manager.py:142-144, command_manager.py:136-142) travel as ONE packed fp64
  vector of ``[sum, count]`` pairs through one RCCL all-reduce(sum) per rollout, issued on a side stream so it
  overlaps the next env steps; means are ``sum(sum) / sum(count)`` (never a mean of means);
* observation-normaliser moments (my_code/models/my_running_standard_scaler.py:57-80) travel as per-rank sums about a
  centre every rank already shares (the running mean): ``[n, sum (x - c), sum (x - c)^2]`` -- additive, so they ride in
  the SAME all-reduce as the episode statistics (``FusedStatistics``: one collective per call), and as well conditioned
  as the Chan partials they are computed from.

Per-rank partials are produced by the HIP kernels ``mdp_stats_masked_sums`` / ``mdp_stats_moments``; everything
here is plumbing around ``torch.distributed`` (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import _lib as L


def shard_range(num_envs_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous env-id range ``[lo, hi)`` owned by ``rank`` (remainder spread over the first ranks)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(num_envs_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def merge_moments(partials: torch.Tensor, unbiased: bool = True) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Chan merge of per-rank ``[n, sum x (D), M2 (D)]`` rows (G, 1 + 2D) fp64 -> (count, mean (D), var (D)).

    ``var`` is the variance of the union the scaler feeds to its running update: ``torch.var`` of the batch
    (``my_running_standard_scaler.py:94-105``), i.e. with Bessel's correction (``unbiased=False``: population)."""
    p = partials.to(torch.float64)
    d = (p.shape[1] - 1) // 2
    n = p[:, 0]
    total = n.sum()
    mean = p[:, 1 : 1 + d].sum(dim=0) / total
    mean_r = p[:, 1 : 1 + d] / n.clamp(min=1.0)[:, None]
    m2 = (p[:, 1 + d :] + n[:, None] * (mean_r - mean[None, :]) ** 2).sum(dim=0)
    return total, mean, m2 / ((total - 1.0).clamp(min=1.0) if unbiased else total)


def running_update(mean, var, count, batch_mean, batch_var, batch_count):
    """my_running_standard_scaler.py:69-80 (parallel-variance update of the running statistics)."""
    delta = batch_mean - mean
    total = count + batch_count
    m2 = var * count + batch_var * batch_count + delta**2 * count * batch_count / total
    return mean + delta * batch_count / total, m2 / total, total


def shifted_sums(partials: torch.Tensor, center: torch.Tensor) -> torch.Tensor:
    """Chan partial rows ``[n, sum x (D), M2 about the row's own mean (D)]`` (C, 1 + 2D) fp64 -> ONE additive row
    ``[n, sum (x - c), sum (x - c)^2]`` about ``center`` (D): exact in real arithmetic (sum (x - c)^2 = M2 + n (mean -
    c)^2 per row), and sums of such rows over ranks are the same quantities of the union."""
    p = partials.to(torch.float64).reshape(-1, partials.shape[-1])
    d = (p.shape[1] - 1) // 2
    n = p[:, 0]
    c = center.to(torch.float64)
    s1 = p[:, 1 : 1 + d] - n[:, None] * c[None, :]
    mean_r = p[:, 1 : 1 + d] / n.clamp(min=1.0)[:, None]
    s2 = p[:, 1 + d :] + n[:, None] * (mean_r - c[None, :]) ** 2
    return torch.cat([n.sum().reshape(1), s1.sum(dim=0), s2.sum(dim=0)])


def chan_row_from_shifted(total: torch.Tensor, center: torch.Tensor) -> torch.Tensor:
    """Inverse map for the merged row: ``[n, sum x, M2 about the batch mean]`` (1, 1 + 2D) -- what ``merge_moments`` and
    ``mdp_scaler_update`` take."""
    d = (total.shape[0] - 1) // 2
    n = total[0]
    s1, s2 = total[1 : 1 + d], total[1 + d :]
    sum_x = s1 + n * center.to(torch.float64)
    m2 = s2 - s1 * s1 / n.clamp(min=1.0)
    return torch.cat([n.reshape(1), sum_x, m2.clamp(min=0.0)]).reshape(1, -1)


class ObsNormalizerStats:
    """Running mean/var of the observation, fed by per-rank HIP partials and merged across ranks."""

    def __init__(self, dim: int, device, group=None):
        self.dim, self.device, self.group = dim, torch.device(device), group
        self.mean = torch.zeros(dim, dtype=torch.float64, device=self.device)
        self.var = torch.ones(dim, dtype=torch.float64, device=self.device)
        self.count = torch.ones((), dtype=torch.float64, device=self.device)
        self.num_chunks = 32
        self._partial = torch.zeros((self.num_chunks, 1 + 2 * dim), dtype=torch.float64, device=self.device)

    def local_partial(self, obs: torch.Tensor) -> torch.Tensor:
        """(num_chunks, 1 + 2D) Chan-mergeable partial rows of this rank's batch (``mdp_stats_moments_chunked``)."""
        obs, ld = L.rows(obs, "obs")
        L.check(L.lib.mdp_stats_moments_chunked(obs.data_ptr(), obs.shape[0], self.dim, ld, self.num_chunks,
                                                self._partial.data_ptr(), L.stream(self.device)))
        return self._partial

    def update(self, obs: torch.Tensor):
        self.merge_partial(self.local_partial(obs))

    def merge_partial(self, part: torch.Tensor):
        """Cross-rank step: this rank's partial rows as ONE additive row about the shared running mean, one
        all-reduce(sum) (RCCL on GPUs, gloo in the CPU tests), then the running update."""
        row = self.local_row(part)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(row, op=dist.ReduceOp.SUM, group=self.group)
        self.fold(row)

    def local_row(self, part: torch.Tensor) -> torch.Tensor:
        """``[n, sum (x - c), sum (x - c)^2]`` of this rank's Chan partial rows, c = the running mean (identical on every
        rank: they have all folded the same merged rows)."""
        return shifted_sums(part.reshape(-1, part.shape[-1]), self.mean)

    def fold(self, row: torch.Tensor):
        """Running update (my_running_standard_scaler.py:69-80) from the merged additive row of ALL ranks."""
        n, mean, var = merge_moments(chan_row_from_shifted(row, self.mean))
        self.mean, self.var, self.count = running_update(self.mean, self.var, self.count, mean, var, n)


class EpisodeStats:
    """Packed ``[sum, count]`` episode statistics with one asynchronous all-reduce per rollout: the four families of
    SURVEY.md 8e(2) -- ``Episode_Reward/*`` (reward_manager.py:115-121), ``Episode_Termination/*``
    (termination_manager.py:142-144: a COUNT over the reset envs), ``Metrics/base_velocity/*`` (command_manager.py:136-142)
    and ``Curriculum/terrain_levels`` (curriculums.py:55: the mean level over ALL envs)."""

    F32, U8, I64 = 0, 1, 2       # element kinds of mdp_stats_masked_sums_typed

    def __init__(self, env, world_size: int = 1, group=None):
        self.env, self.world, self.group = env, world_size, group
        self.device = torch.device(env.device)
        rm, tm = env.reward_manager, env.termination_manager
        self.names = ["Episode_Reward/" + n for n in rm.active_terms]
        # (buffer, rows, element kind, follows the reset mask)
        self._sources = [(rm._episode_sums_block, len(rm.active_terms), self.F32, True)]
        cmd = env.command_manager.get_term("base_velocity") if "base_velocity" in env.command_manager.active_terms else None
        if cmd is not None:
            for k, v in cmd.metrics.items():
                self.names.append("Metrics/base_velocity/" + k)
                self._sources.append((v, 1, self.F32, True))
        term_block = getattr(tm, "_term_dones_block", None)
        if term_block is not None and len(tm.active_terms) > 0:
            self.names += ["Episode_Termination/" + n for n in tm.active_terms]
            self._sources.append((term_block, len(tm.active_terms), self.U8, True))
        terrain = getattr(getattr(env, "scene", None), "terrain", None)
        if terrain is not None and hasattr(terrain, "device_state"):
            levels = terrain.device_state(env.device)["terrain_levels"]
            self.names.append("Curriculum/terrain_levels")
            self._sources.append((levels, 1, self.I64, False))
        self.payload = torch.zeros(2 * len(self.names), dtype=torch.float64, device=self.device)
        self._table = {}            # mdp_stats_source_t records of _sources, with / without a mask (the buffers are persistent)
        self._separate = os.environ.get("MDP_STATS_SEPARATE") == "1"   # A/B knob: one launch per family, side stream even alone
        self._stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._work = None
        self._ready = torch.cuda.Event() if self._stream is not None else None

    def fill(self, out: torch.Tensor, mask: torch.Tensor | None, stream) -> None:
        """This rank's ``[sum, count]`` pairs into ``out`` (2 * len(names) doubles): ONE launch on ``stream`` for all
        families (``mdp_stats_masked_sums_multi``; one ``mdp_stats_masked_sums_typed`` launch per family when there are
        more families than its table holds -- the same sums either way)."""
        if not self._separate and len(self._sources) <= L.MDP_STATS_MAX_SOURCES:
            key = mask is not None          # without a mask every family sums over all envs
            if self._table.get(key) is None:
                table = (L.MdpStatsSource * len(self._sources))()
                for t, (src, rows, kind, masked) in zip(table, self._sources):
                    t.values, t.elem_kind, t.rows, t.masked = src.data_ptr(), kind, rows, int(masked and key)
                self._table[key] = table
            L.check(L.lib.mdp_stats_masked_sums_multi(self._table[key], len(self._sources), self.env.num_envs, L.ptr(mask),
                                                      out.data_ptr(), stream))
            return
        off = 0
        for src, rows, kind, masked in self._sources:
            L.check(L.lib.mdp_stats_masked_sums_typed(src.data_ptr(), kind, rows, self.env.num_envs,
                                                      L.ptr(mask) if masked else None, out[off:].data_ptr(), stream))
            off += 2 * rows

    def reduce(self, mask: torch.Tensor | None = None, force_collective: bool = False):
        """Per-rank partial sums and one all-reduce(sum) of the packed vector, in order on the current stream."""
        self.fill(self.payload, mask, L.stream(self.device))
        work = allreduce_payload(self.payload, self.world, self.group, force=force_collective)
        if work is not None:
            work.wait()

    def reduce_async(self, mask: torch.Tensor | None = None, force_collective: bool = False):
        """Per-rank partial sums on the CURRENT stream, in order behind the step (one launch), then one asynchronous
        all-reduce(sum) of the packed vector: RCCL's own stream takes over from the current one through an event, the
        current stream goes on with the next rollout and only waits for the collective when the payload is needed again
        (the next call, or ``result()``).  Up to round 3's last day the sums ran on a side stream of their own: one more
        cross-queue hand-over (~25 us on this stack) in front of the collective, at the end of every timed region, and
        the sums read buffers the step's stream was already writing again.  ``MDP_STATS_SEPARATE=1`` restores that form
        (A/B knob)."""
        if self._separate:
            return self._reduce_async_side_stream(mask, force_collective)
        if self._work is not None:          # the previous collective still owns the payload (device-side wait)
            self._work.wait()
            self._work = None
        self.fill(self.payload, mask, L.stream(self.device))
        self._work = allreduce_payload(self.payload, self.world, self.group, force=force_collective)

    def _reduce_async_side_stream(self, mask, force_collective):
        cur = torch.cuda.current_stream(self.device)
        self._ready.record(cur)
        with torch.cuda.stream(self._stream):
            self._stream.wait_event(self._ready)
            self.fill(self.payload, mask, self._stream.cuda_stream)
            self._work = allreduce_payload(self.payload, self.world, self.group, force=force_collective)
            if self._work is not None:
                # device-side wait (RCCL: the side stream waits for the collective's stream, the host does not block):
                # the next rollout's partial sums must not overwrite the payload while it is being reduced
                self._work.wait()

    def result(self) -> dict[str, float]:
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self._stream is not None:
            self._stream.synchronize()
        return finalize_stats(self.names, self.payload.cpu(), self.env.max_episode_length_s)


class FusedStatistics:
    """ONE all-reduce(sum) for everything that crosses ranks in a call: the observation normaliser's
    additive row ``[n, sum (x - c), sum (x - c)^2]`` (1 + 2D doubles) in front of the episode statistics' ``[sum, count]``
    pairs, one packed fp64 vector.  ``step(obs, mask)``: this rank's partials -> the collective -> the normaliser's
    running update from the merged row (``RunningStandardScaler.apply_merged_row``); the statistics stay in
    ``payload[stats_offset:]`` for ``result()``."""

    def __init__(self, episode: "EpisodeStats | None", scaler=None, world_size: int = 1, group=None):
        self.episode, self.scaler, self.world, self.group = episode, scaler, world_size, group
        dev = episode.device if episode is not None else scaler.device
        self.device = torch.device(dev)
        self.stats_offset = (1 + 2 * scaler.size) if scaler is not None else 0
        n_stats = 2 * len(episode.names) if episode is not None else 0
        self.payload = torch.zeros(self.stats_offset + n_stats, dtype=torch.float64, device=self.device)

    def step(self, obs: torch.Tensor | None = None, mask: torch.Tensor | None = None, force_collective: bool = False):
        stream = L.stream(self.device)
        if self.scaler is not None:
            if obs is None:
                raise ValueError("the fused payload carries the normaliser's row: pass the observation batch")
            self.payload[: self.stats_offset].copy_(self.scaler.local_row(obs))
        if self.episode is not None:
            self.episode.fill(self.payload[self.stats_offset :], mask, stream)
        work = allreduce_payload(self.payload, self.world, self.group, force=force_collective)
        if work is not None:
            work.wait()          # device-side for RCCL: the current stream waits for the collective's stream
        if self.scaler is not None:
            self.scaler.apply_merged_row(self.payload[: self.stats_offset])

    def result(self) -> dict[str, float]:
        if self.episode is None:
            return {}
        return finalize_stats(self.episode.names, self.payload[self.stats_offset :].cpu(), self.episode.env.max_episode_length_s)


def allreduce_payload(payload: torch.Tensor, world: int, group=None, force: bool = False):
    """One asynchronous all-reduce(sum) of the packed ``[sum, count]`` vector; returns the work handle (None when
    there is a single rank).  ``force``: issue the collective even with one rank (a one-GPU box runs the RCCL branch --
    communicator, collective stream, ``Work.wait()`` -- once before an 8-GPU node does; tests/test_gpu_rccl.py)."""
    if world > 1 or (force and dist.is_available() and dist.is_initialized()):
        if payload.is_cuda and dist.get_backend(group) == "gloo" and os.environ.get("MDP_GLOO_DEVICE_PATH") != "1":
            # gloo is the CPU transport (tests, one-box rehearsals).  Its device-tensor path (ProcessGroupGloo stages the
            # tensor through pinned host memory on a pool stream and synchronises that stream from its worker thread
            # while this thread blocks in Work.wait()) stalls in the two-rank one-GPU rehearsal: round-2 diagnosis
            # (tools/rehearse_two_ranks.sh, MDP_GLOO_DEVICE_PATH=1 + watchdog) -- the first timed region passes, in the
            # second both ranks sit in Work.wait() of the same collective for good; with the payload staged through the
            # host by this thread the same rehearsal completes.  The stall is inside gloo's staging, which the RCCL
            # branch below never enters (one collective kernel on RCCL's stream, Work.wait() = a device-side event wait).
            host = payload.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            payload.copy_(host)
            return None
        return dist.all_reduce(payload, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return None


def finalize_stats(names: list[str], payload: torch.Tensor, max_episode_length_s: float) -> dict[str, float]:
    """``sum(sum)/sum(count)``; episodic reward sums are divided by the episode length in seconds like
    reward_manager.py:119; ``Episode_Termination/*`` is the summed count itself."""
    out = {}
    for i, name in enumerate(names):
        s, c = float(payload[2 * i]), float(payload[2 * i + 1])
        if name.startswith("Episode_Termination/"):       # termination_manager.py:142-144: count_nonzero, not a mean
            out[name] = s
            continue
        mean = s / c if c > 0 else 0.0
        out[name] = mean / max_episode_length_s if name.startswith("Episode_Reward/") else mean
    return out
