#!/usr/bin/env python3
"""Micro-benchmark of mdp_raycast on the benchmark terrain (GPU box): isolates fixed per-ray cost, locality and
structure effects.  python tools/bench_raycast.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from isaaclab_amd.terrains import SyntheticTerrain, SyntheticTerrainCfg
from isaaclab_amd.utils.warp import convert_to_warp_mesh, raycast_mesh
from isaaclab_amd.terrains.synthetic import make_plane


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def grid_rays(pos, yaw):
    xs = np.arange(-0.8, 0.8 + 1e-9, 0.1)
    ys = np.arange(-0.5, 0.5 + 1e-9, 0.1)
    gx, gy = np.meshgrid(xs, ys, indexing="xy")
    loc = np.stack([gx.ravel(), gy.ravel()], 1)
    c, s = np.cos(yaw)[:, None], np.sin(yaw)[:, None]
    wx = pos[:, None, 0] + c * loc[None, :, 0] - s * loc[None, :, 1]
    wy = pos[:, None, 1] + s * loc[None, :, 0] + c * loc[None, :, 1]
    st = np.stack([wx, wy, np.full_like(wx, 20.0) + pos[:, None, 2]], -1).astype(np.float32)
    d = np.zeros_like(st)
    d[..., 2] = -1.0
    return st.reshape(-1, 3), d.reshape(-1, 3)


def main():
    dev = "cuda:0"
    n = 4096
    t0 = time.time()
    terrain = SyntheticTerrain(SyntheticTerrainCfg(), n)
    rng = np.random.default_rng(0)
    pos = terrain.env_origins + np.concatenate([rng.uniform(-0.5, 0.5, (n, 2)), np.zeros((n, 1))], 1)
    yaw = rng.uniform(-np.pi, np.pi, n)
    st, d = grid_rays(pos, yaw)
    print(f"terrain+rays {time.time() - t0:.2f}s  rays={st.shape[0]}")
    for accel in ("grid", "bvh"):
        t0 = time.time()
        mesh = convert_to_warp_mesh(terrain.vertices, terrain.triangles, dev, accel=accel)
        print(accel, "build %.2fs" % (time.time() - t0), mesh.info())
        S, D = torch.from_numpy(st).to(dev), torch.from_numpy(d).to(dev)
        print(f"  {accel}: bench rays            {timeit(lambda: raycast_mesh(S, D, mesh)):9.1f} us")
        # all envs at one place: cache-hot geometry
        st1, d1 = grid_rays(np.repeat(pos[:1], n, 0), np.repeat(yaw[:1], n))
        S1, D1 = torch.from_numpy(st1).to(dev), torch.from_numpy(d1).to(dev)
        print(f"  {accel}: same pose x4096       {timeit(lambda: raycast_mesh(S1, D1, mesh)):9.1f} us")
        for frac in (4, 16):
            Sq, Dq = S[: S.shape[0] // frac].contiguous(), D[: D.shape[0] // frac].contiguous()
            print(f"  {accel}: 1/{frac} of the rays        {timeit(lambda: raycast_mesh(Sq, Dq, mesh)):9.1f} us")
        # per-tile-type timing: envs grouped by terrain column (type)
        types = terrain.terrain_types
        for lo, hi, name in ((0, 4, "stairs"), (4, 8, "inv stairs"), (8, 12, "boxes"), (12, 16, "rough hf"), (16, 20, "slopes")):
            sel = np.nonzero((types >= lo) & (types < hi))[0]
            stt, dt = grid_rays(pos[sel], yaw[sel])
            St, Dt = torch.from_numpy(stt).to(dev), torch.from_numpy(dt).to(dev)
            print(f"  {accel}: {name:<11} {len(sel):5d} envs  {timeit(lambda: raycast_mesh(St, Dt, mesh)):9.1f} us")
    v, f = make_plane((200.0, 200.0))
    mesh = convert_to_warp_mesh(v, f, dev)
    print(f"  plane (2 tris): bench rays    {timeit(lambda: raycast_mesh(S, D, mesh)):9.1f} us", mesh.info())
    out = torch.empty_like(S)
    print(f"  copy of the ray array         {timeit(lambda: out.copy_(S)):9.1f} us")


if __name__ == "__main__":
    main()
