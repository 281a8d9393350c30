#!/usr/bin/env python3
"""Wide parity sweep of the ray cast on the benchmark terrain (GPU box): what tests/test_gpu_full_terrain.py checks on a
>= 20 k-ray slice, here for EVERY ray of 4096 envs x 187 rays over many seeds of sensor poses, plus tilted rays through
the generic cast.  The answer is oracle/raycast_ref.c (independent restatement, no shared source with the product).

    python tools/soak_raycast.py [--seeds 40] [--tilted 200000]

Prints one line per seed and a total; exit status 1 on the first bit that differs."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from isaaclab_amd import _lib as L
from isaaclab_amd.envs.synthetic import make_rough_terrain
from isaaclab_amd.utils.warp import convert_to_warp_mesh, raycast_mesh
from oracle import mdp_np as P
from oracle import raycast_np as RC


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--tilted", type=int, default=200000)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n = 4096
    terrain = make_rough_terrain(n)
    mesh = convert_to_warp_mesh(terrain.vertices, terrain.triangles, str(dev), accel="grid")
    soup = RC.tri_soup(terrain.vertices, terrain.triangles)
    caster = RC.GridCaster(soup, cell=0.25)
    starts, dirs = P.grid_pattern(0.1, [1.6, 1.0])
    starts = starts + np.array([0, 0, 20.0], np.float32)
    R = starts.shape[0]
    d_starts, d_dirs = torch.from_numpy(starts).to(dev), torch.from_numpy(dirs).to(dev)
    half_x, half_y = 10 * 8.0 / 2 + 20.0, 20 * 8.0 / 2 + 20.0
    total_v = total_t = miss_v = miss_t = 0
    t0 = time.time()
    for seed in range(args.seeds):
        rng = np.random.default_rng(1000 + seed)
        # sensor poses anywhere over the terrain and a little beyond its outer edge, every yaw, plus axis-aligned ones
        pos = np.stack([rng.uniform(-half_x - 1.0, half_x + 1.0, n), rng.uniform(-half_y - 1.0, half_y + 1.0, n),
                        rng.uniform(0.3, 1.5, n)], 1).astype(np.float32)
        k = n // 4
        pos[:k] = (terrain.env_origins[:k] + rng.normal(0, 1.5, (k, 3)) * np.array([1, 1, 0.05]) + np.array([0, 0, 0.6])).astype(np.float32)
        yaw = rng.uniform(-np.pi, np.pi, n)
        yaw[:: 16] = rng.integers(-2, 3, len(yaw[:: 16])) * (np.pi / 2)
        quat = np.stack([np.cos(yaw / 2), 0 * yaw, 0 * yaw, np.sin(yaw / 2)], 1)
        pose = np.concatenate([pos, quat], 1).astype(np.float32)
        d_pose = torch.from_numpy(pose).to(dev)
        pos_w, quat_w = torch.zeros(n, 3, device=dev), torch.zeros(n, 4, device=dev)
        hits = torch.full((n, R, 3), -7.0, device=dev)
        L.check(L.lib.mdp_raycaster_update(mesh.handle, n, R, d_pose.data_ptr(), 7, None, d_starts.data_ptr(),
                                           d_dirs.data_ptr(), 1, 1, 1e6, pos_w.data_ptr(), quat_w.data_ptr(),
                                           hits.data_ptr(), None, None, None, None, 0, 0.0, 0.0, L.stream(dev)))
        torch.cuda.synchronize()
        got = hits.cpu().numpy()
        sw, dw = P.raycaster_rays(pose[:, :3], pose[:, 3:], starts, dirs, True)
        fin = np.isfinite(got).all(-1)
        # the kernel's own ray origins (a vertical ray keeps its xy; numpy's trig differs in the last bit for some yaws)
        if np.abs(got[fin][:, :2] - sw[fin][:, :2]).max() > 2e-5:
            print(f"seed {seed}: ray origins differ by more than rounding"); return 1
        sw = sw.copy()
        sw[fin, :2] = got[fin][:, :2]
        exp_hits, _, _, exp_face = caster.cast(sw.reshape(-1, 3), dw.reshape(-1, 3))
        bad_v = int((got.reshape(-1, 3).view(np.uint32) != exp_hits.view(np.uint32)).any(-1).sum())
        # rays that missed on the GPU kept the numpy origin: the oracle must miss there too
        total_v += n * R
        miss_v += int((exp_face < 0).sum())
        # tilted rays through the generic cast (DDA over the grid): hits, distances, normals and face ids
        m = args.tilted
        s = np.stack([rng.uniform(-half_x - 2, half_x + 2, m), rng.uniform(-half_y - 2, half_y + 2, m), rng.uniform(0.5, 6.0, m)], 1).astype(np.float32)
        d = np.stack([rng.normal(0, 0.5, m), rng.normal(0, 0.5, m), -np.abs(rng.normal(1.0, 0.3, m)) - 0.05], 1)
        d[: m // 8, 2] *= -0.2                                    # some upward rays (misses or the underside of overhangs)
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        e_hits, e_dist, e_normal, e_face = caster.cast(s, d, 50.0)
        g_hits, g_dist, g_normal, g_face = raycast_mesh(torch.from_numpy(s).to(dev), torch.from_numpy(d).to(dev), mesh,
                                                        50.0, True, True, True)
        torch.cuda.synchronize()
        bad_t = int(((g_face.cpu().numpy() != e_face) | (g_hits.cpu().numpy().view(np.uint32) != e_hits.view(np.uint32)).any(-1)
                     | (g_dist.cpu().numpy().view(np.uint32) != e_dist.view(np.uint32))
                     | (g_normal.cpu().numpy().view(np.uint32) != e_normal.view(np.uint32)).any(-1)).sum())
        total_t += m
        miss_t += int((e_face < 0).sum())
        print(f"seed {seed:3d}: vertical {n * R} rays, {int((exp_face < 0).sum())} misses, {bad_v} differ | "
              f"tilted {m} rays, {int((e_face < 0).sum())} misses, {bad_t} differ   [{time.time() - t0:.0f} s]", flush=True)
        if bad_v or bad_t:
            return 1
    print(f"TOTAL: {total_v} vertical rays ({miss_v} misses) and {total_t} tilted rays ({miss_t} misses) on "
          f"{terrain.triangles.shape[0]} triangles: every hit point, distance, normal and face id bit-identical to oracle/raycast_ref.c")
    return 0


if __name__ == "__main__":
    sys.exit(main())
