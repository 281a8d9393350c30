#!/usr/bin/env python3
"""Wide parity sweep of the env step (GPU box): what tests/test_gpu_env_step.py checks for one seed per task, here over
many seeds of frames / actions / injected random numbers, for the four velocity tasks, without resets and with the
masked reset inside the step.  HIP path (the single post-physics launch where the task allows it) against the numpy
oracle on the same recording; the test suite's criterion (masks bit-exact; rewards rtol 1e-5 + 2e-6; observations
rtol 1e-5 + 1e-5) is applied to every element and the largest deviations are printed.

    python tools/soak_env_step.py [--seeds 12] [--envs 256] [--steps 12]

Exit status 1 if any element is out of tolerance."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from isaaclab_amd.envs.synthetic import make_velocity_env
from oracle.harness import replay_oracle


def run(name, n, steps, seed, reset):
    kw = dict(small_terrain=True, mesh_accel="grid") if "rough" in name else {}
    env, rec = make_velocity_env(name, num_envs=n, device="cuda:0", steps=steps, seed=seed, **kw)
    env.observation_manager.persistent_output = True
    start = None
    if reset:
        env.auto_reset = "masked"
        g = torch.Generator().manual_seed(seed)
        start = env.max_episode_length - torch.randint(1, steps + 3, (n,), generator=g)
        env.episode_length_buf.copy_(start)
    outs = []
    for t in range(steps):
        rec.feed(t)
        obs, rew, term, trunc, _ = env.step(rec.actions[t].to("cuda:0"))
        outs.append((obs["policy"].cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy(), trunc.cpu().numpy()))
    torch.cuda.synchronize()
    if reset:
        ref = replay_oracle(rec, steps, auto_reset=True, episode_length0=start.numpy())
    else:
        ref = replay_oracle(rec, steps)
    return outs, ref, getattr(env, "_single_launch_steps", 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=12)
    ap.add_argument("--envs", type=int, default=256)
    ap.add_argument("--steps", type=int, default=12)
    args = ap.parse_args()
    bad_total = 0
    t0 = time.time()
    print(f"{'task':<16}{'reset':<8}{'seeds':>6}{'elements':>12}{'mask bits':>11}{'resets':>8}{'obs max abs':>13}{'obs max rel*':>14}"
          f"{'rew max abs':>13}{'rew max rel*':>14}{'out of tol':>12}{'single':>8}")
    for name in ("anymal_d_flat", "anymal_d_rough", "anymal_c_rough", "h1_rough"):
        for reset in (False, True):
            n_el = n_mask = n_bad = n_reset = single = 0
            oa = orr = ra = rr = 0.0
            for seed in range(args.seeds):
                outs, ref, sl = run(name, args.envs, args.steps, 500 + seed, reset)
                single += sl
                for got, exp in zip(outs, ref):
                    n_bad += int((got[2] != exp[2]).sum() + (got[3] != exp[3]).sum())
                    n_mask += got[2].size + got[3].size
                    n_reset += int((exp[2] | exp[3]).sum())
                    for k, atol in ((0, 1e-5), (1, 2e-6)):
                        g, e = got[k].astype(np.float64), np.asarray(exp[k], np.float64)
                        assert g.shape == e.shape
                        d = np.abs(g - e)
                        n_bad += int((d > atol + 1e-5 * np.abs(e)).sum()) + int((np.isfinite(g) != np.isfinite(e)).sum())
                        n_el += g.size
                        d = np.where(np.isfinite(d), d, 0.0)
                        rel = d / np.maximum(np.abs(e), 1.0)         # * relative to max(|expected|, 1)
                        if k == 0:
                            oa, orr = max(oa, d.max()), max(orr, rel.max())
                        else:
                            ra, rr = max(ra, d.max()), max(rr, rel.max())
            bad_total += n_bad
            print(f"{name:<16}{'masked' if reset else 'off':<8}{args.seeds:>6}{n_el:>12}{n_mask:>11}{n_reset:>8}{oa:>13.2e}{orr:>14.2e}"
                  f"{ra:>13.2e}{rr:>14.2e}{n_bad:>12}{single:>8}   [{time.time() - t0:.0f} s]", flush=True)
    print("TOTAL out of tolerance:", bad_total)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
