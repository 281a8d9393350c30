#!/usr/bin/env python3
"""Observation standardisation + rollout-storage write at the benchmark shape (4096 x 235):
mdp_scaler_apply straight into the storage slot vs the reference's torch expression + copy, both replayed from a
hipGraph of 20 repetitions (HIP events; kernel time + graph gaps, no Python)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isaaclab_amd.rl import RunningStandardScaler


def graph_time(fn, reps=20, replays=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(replays):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * replays)


def main():
    n, o, T = 4096, 235, 24
    dev = "cuda:0"
    obs = torch.randn(n, o, device=dev) * 3 + 1
    storage = torch.zeros(T, n, o, device=dev)
    sc = RunningStandardScaler(o, device=dev)
    sc(obs, train=True)
    mean, var, eps, clip = sc.running_mean, sc.running_variance, sc.epsilon, sc.clip_threshold

    def hip():
        sc(obs, out=storage[3])

    def ref():
        storage[3].copy_(torch.clamp((obs - mean.float()) / (torch.sqrt(var.float()) + eps), min=-clip, max=clip))

    def hip_train():
        sc(obs, train=True, out=storage[3])

    t_hip, t_ref, t_train = graph_time(hip), graph_time(ref), graph_time(hip_train)
    bytes_ = 2 * n * o * 4
    print(f"standardise + clip + storage write, {n} x {o}: HIP {t_hip:.2f} us ({bytes_ / t_hip / 1e3:.0f} GB/s algorithmic), "
          f"torch expression + copy {t_ref:.2f} us; with the running-statistics update {t_train:.2f} us")


if __name__ == "__main__":
    main()
