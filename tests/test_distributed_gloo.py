"""Multi-process path on CPU: world_size-2 ``gloo`` ranks run the same host code the GPU ranks run over
RCCL -- env sharding, the all-gather + Chan merge of the observation-normaliser partials, the packed [sum, count]
episode-statistics all-reduce and the Shadow-hand ``consecutive_successes`` partial sums.  The per-rank partials come
from numpy here (on a GPU rank they come from mdp_stats_moments / mdp_stats_masked_sums / mdp_inhand_rewards, which
tests/test_gpu_scene.py and tests/test_inhand.py pin); what is checked is that the merged result is independent of the
shard count: equal to the single-process value."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 2
N, D, K = 1003, 48, 5     # envs (not divisible by the world size on purpose), obs width, logged keys


def _data():
    rng = np.random.RandomState(77)
    obs = (rng.randn(3, N, D) * 2.5 + 0.7).astype(np.float32)       # three rollout steps
    sums = rng.randn(K, N).astype(np.float32)
    mask = rng.rand(N) < 0.2
    succ = rng.randint(0, 6, N).astype(np.float32)
    resets = rng.rand(N) < 0.15
    return obs, sums, mask, succ, resets


def _term_data():
    rng = np.random.RandomState(78)
    return rng.rand(2, N) < 0.3, rng.randint(0, 10, N).astype(np.int64)


def _extra_batch():
    return (np.random.RandomState(79).randn(N, D) * 0.8 - 1.1).astype(np.float32)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from isaaclab_amd import distributed as D_

    obs, sums, mask, succ, resets = _data()
    lo, hi = D_.shard_range(N, rank, WORLD)
    # (1) observation normaliser: per-rank [n, sum x, M2 about the rank mean] -> all-gather -> Chan merge -> running update
    norm = D_.ObsNormalizerStats(D, "cpu")
    for t in range(obs.shape[0]):
        x = obs[t, lo:hi].astype(np.float64)
        part = np.concatenate([[hi - lo], x.sum(0), ((x - x.mean(0)) ** 2).sum(0)])
        norm.merge_partial(torch.from_numpy(part))
    # (2) episode statistics, all four families of SURVEY 8e(2) as packed [sum, count] pairs -- and in the SAME vector,
    # in front of them, the normaliser's additive row of one more batch: ONE all-reduce for everything (FusedStatistics'
    # layout; the per-rank partials are numpy here, HIP kernels on a GPU rank)
    names = ([f"Episode_Reward/r{k}" for k in range(K - 1)] + ["Metrics/base_velocity/error_vel_xy"]
             + ["Episode_Termination/time_out", "Episode_Termination/base_contact", "Curriculum/terrain_levels"])
    term_flags, levels = _term_data()
    m = mask[lo:hi]
    pairs = []
    for k in range(K):
        pairs += [float(sums[k, lo:hi].astype(np.float64)[m].sum()), float(m.sum())]
    for t in range(2):                                    # count_nonzero over the reset envs (termination_manager.py:142-144)
        pairs += [float(term_flags[t, lo:hi][m].sum()), float(m.sum())]
    pairs += [float(levels[lo:hi].sum()), float(hi - lo)]  # mean level over ALL envs (curriculums.py:55)
    x = _extra_batch()[lo:hi].astype(np.float64)
    part = torch.from_numpy(np.concatenate([[hi - lo], x.sum(0), ((x - x.mean(0)) ** 2).sum(0)]))
    fused = torch.cat([norm.local_row(part), torch.tensor(pairs, dtype=torch.float64)])
    work = D_.allreduce_payload(fused, WORLD)
    work.wait()
    norm.fold(fused[: 1 + 2 * D])
    payload = fused[1 + 2 * D :]
    stats = D_.finalize_stats(names, payload, 20.0)
    # (3) Shadow hand: [sum resets, sum successes*resets] (inhand_manipulation_env.py:424-431)
    partials = torch.tensor([float(resets[lo:hi].sum()), float((succ[lo:hi] * resets[lo:hi]).sum())], dtype=torch.float64)
    dist.all_reduce(partials)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), mean=norm.mean.numpy(), var=norm.var.numpy(),
             count=norm.count.numpy(), stats=np.array([stats[n] for n in names]), partials=partials.numpy(),
             shard=np.array([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_every_env():
    sys.path.insert(0, ROOT)
    from isaaclab_amd.distributed import shard_range

    for total, world in ((4096, 8), (1003, 2), (7, 8), (0, 3)):
        edges = [shard_range(total, r, world) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
        sizes = [hi - lo for lo, hi in edges]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_two_rank_gloo_matches_single_process(tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.start_processes(_rank_main, args=(port, str(tmp_path)), nprocs=WORLD, join=True, start_method="spawn")
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(WORLD)]
    obs, sums, mask, succ, resets = _data()
    # single-process reference of the same running update (my_running_standard_scaler.py:57-80: count starts at 1,
    # mean 0, var 1; batch variance is torch.var of the merged batch, Bessel-corrected)
    mean, var, count = np.zeros(D), np.ones(D), 1.0
    for x in [obs[t].astype(np.float64) for t in range(obs.shape[0])] + [_extra_batch().astype(np.float64)]:
        bm, bv, bn = x.mean(0), x.var(0, ddof=1), float(N)
        delta, tot = bm - mean, count + bn
        m2 = var * count + bv * bn + delta ** 2 * count * bn / tot
        mean, var, count = mean + delta * bn / tot, m2 / tot, tot
    for k in range(WORLD):
        np.testing.assert_allclose(r[k]["mean"], mean, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(r[k]["var"], var, rtol=1e-11, atol=1e-12)
        assert float(r[k]["count"]) == count
    exp = [sums[k].astype(np.float64)[mask].sum() / mask.sum() for k in range(K)]
    exp = [e / 20.0 for e in exp[:-1]] + [exp[-1]]     # episodic reward sums are divided by the episode length in s
    term_flags, levels = _term_data()
    exp += [float(term_flags[0][mask].sum()), float(term_flags[1][mask].sum()), float(levels.mean())]
    for k in range(WORLD):
        np.testing.assert_allclose(r[k]["stats"], exp, rtol=1e-12)
        np.testing.assert_array_equal(r[k]["partials"], [resets.sum(), (succ * resets).sum()])
    assert r[0]["shard"][1] == r[1]["shard"][0] and r[1]["shard"][1] == N
