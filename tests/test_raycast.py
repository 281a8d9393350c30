"""Ray caster: C oracle self-checks on CPU and bit-exact HIP-vs-oracle parity on the GPU.

Parity note ("parity unpinned" against the reference): the reference delegates the closest-hit query to
NVIDIA Warp's ``wp.mesh_query_ray`` (LAB/utils/warp/kernels.py:14-75; warp-lang is a pip dependency that
is absent from this image and from /root/reference).  The oracle restates Warp's published algorithm
(watertight Woop/Benthin/Wald ray-triangle test, closest hit) and is pinned by analytic known answers
(plane / ramp / box hits computed by hand below) instead of reference outputs.  The HIP kernels share
the arithmetic header with the oracle, so HIP-vs-oracle is bit-exact: hit point, distance, normal, face id
(ties resolved to the smallest face id) for the grid, the BVH and the brute-force traversal."""
import numpy as np
import pytest

from conftest import to_dev, to_np
from oracle import raycast_np as RC


def plane(z=0.0, half=50.0):
    v = np.array([[-half, -half, z], [half, -half, z], [half, half, z], [-half, half, z]], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    return v, f


def heightfield(seed=0, n=48, scale=0.25, amp=0.4):
    rng = np.random.default_rng(seed)
    z = (rng.random((n, n)) * amp).astype(np.float32)
    xs = (np.arange(n) * scale - n * scale / 2).astype(np.float32)
    xx, yy = np.meshgrid(xs, xs, indexing="ij")
    v = np.stack([xx.ravel(), yy.ravel(), z.ravel()], axis=1).astype(np.float32)
    i, j = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    a = (i * n + j).ravel()
    f = np.concatenate([np.stack([a, a + n + 1, a + 1], 1), np.stack([a, a + n, a + n + 1], 1)]).astype(np.int32)
    return v, f


def rays_down(seed, n, extent, z0=20.0, tilt=0.0):
    rng = np.random.default_rng(seed)
    s = np.zeros((n, 3), np.float32)
    s[:, :2] = (rng.random((n, 2)) * 2 - 1) * extent
    s[:, 2] = z0
    d = np.zeros((n, 3), np.float32)
    d[:, 2] = -1.0
    if tilt:
        d[:, :2] = (rng.standard_normal((n, 2)) * tilt).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
    return s, d.astype(np.float32)


# ---- CPU: oracle known answers -----------------------------------------------------------------------
def test_oracle_plane_known_answers():
    v, f = plane(0.5)
    soup = RC.tri_soup(v, f)
    s = np.array([[0.3, -0.2, 20.0], [1.0, 2.0, 3.0], [0.0, 0.0, 0.0], [100.0, 0.0, 5.0]], np.float32)
    d = np.array([[0, 0, -1.0], [0, 0, -1.0], [0, 0, -1.0], [0, 0, -1.0]], np.float32)
    hits, dist, normal, face = RC.raycast_brute(soup, s, d, 1e6)
    np.testing.assert_allclose(hits[0], [0.3, -0.2, 0.5], atol=1e-6)
    np.testing.assert_allclose(dist[:2], [19.5, 2.5], atol=1e-6)
    np.testing.assert_allclose(np.abs(normal[0]), [0, 0, 1], atol=1e-6)
    # below the plane looking down, and outside the plane: misses are inf / -1 (ops.py:46-58)
    assert np.all(np.isinf(hits[2])) and np.isinf(dist[2]) and face[2] == -1
    assert np.all(np.isinf(hits[3])) and face[3] == -1
    # max distance cuts the hit
    hits2, _, _, face2 = RC.raycast_brute(soup, s[:1], d[:1], 10.0)
    assert face2[0] == -1 and np.all(np.isinf(hits2[0]))


def test_oracle_ramp_and_tilted_ray():
    # ramp z = 0.5 * x over [-4, 4]^2 ; ray from (1, 0, 10) straight down hits z = 0.5
    v = np.array([[-4, -4, -2], [4, -4, 2], [4, 4, 2], [-4, 4, -2]], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    soup = RC.tri_soup(v, f)
    hits, dist, normal, _ = RC.raycast_brute(soup, np.array([[1, 0, 10.0]], np.float32), np.array([[0, 0, -1.0]], np.float32))
    np.testing.assert_allclose(hits[0], [1, 0, 0.5], atol=1e-6)
    np.testing.assert_allclose(dist[0], 9.5, atol=1e-6)
    n = np.array([-0.5, 0, 1.0]) / np.sqrt(1.25)
    np.testing.assert_allclose(np.abs(normal[0] @ n), 1.0, atol=1e-6)
    # 45-degree ray in the xz plane from (-3, 0, 3): x = -3 + s, z = 3 - s = 0.5 x -> s = 3, hit (0, 0, 0)
    dd = np.array([[1, 0, -1.0]], np.float32) / np.sqrt(2).astype(np.float32)
    hits, dist, _, _ = RC.raycast_brute(soup, np.array([[-3, 0, 3.0]], np.float32), dd.astype(np.float32))
    np.testing.assert_allclose(hits[0], [0, 0, 0], atol=2e-6)
    np.testing.assert_allclose(dist[0], 3 * np.sqrt(2), rtol=1e-6)


def test_oracle_grid_equals_brute():
    v, f = heightfield(1)
    soup = RC.tri_soup(v, f)
    g = RC.GridCaster(soup)
    for tilt in (0.0, 0.3):
        s, d = rays_down(2, 3000, 6.5, tilt=tilt)
        a = RC.raycast_brute(soup, s, d)
        b = g.cast(s, d)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    # a shared edge / vertex hit resolves to the smallest face id in both
    s = v[f[:, 0]][:200].copy()
    s[:, 2] = 10.0
    d = np.tile(np.array([0, 0, -1.0], np.float32), (200, 1))
    a, b = RC.raycast_brute(soup, s, d), g.cast(s, d)
    np.testing.assert_array_equal(a[3], b[3])
    np.testing.assert_array_equal(a[0], b[0])


def _moller_trumbore_f64(v, f, s, d):
    """Independent float64 closest-hit brute force (Moller-Trumbore, two-sided): a second opinion on the restated
    watertight test -- a different formula evaluated in a different precision by different code."""
    tri = v[f].astype(np.float64)
    a, e1, e2 = tri[:, 0], tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    best_t = np.full(len(s), np.inf)
    best_f = np.full(len(s), -1, np.int64)
    for i in range(len(s)):
        o, dd = s[i].astype(np.float64), d[i].astype(np.float64)
        p = np.cross(dd, e2)
        det = (e1 * p).sum(1)
        ok = np.abs(det) > 1e-14
        inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
        tv = o - a
        u = (tv * p).sum(1) * inv
        q = np.cross(tv, e1)
        w = (q * dd).sum(1) * inv
        t = (q * e2).sum(1) * inv
        eps = 1e-9
        hit = ok & (u >= -eps) & (w >= -eps) & (u + w <= 1 + eps) & (t >= 0)
        if hit.any():
            t = np.where(hit, t, np.inf)
            k = int(np.argmin(t))
            best_t[i], best_f[i] = t[k], k
    return best_t, best_f


def test_oracle_vs_independent_float64_brute_force():
    v, f = heightfield(3, n=24)
    soup = RC.tri_soup(v, f)
    for tilt in (0.0, 0.08):
        s, d = rays_down(5, 800, 3.2, tilt=tilt)
        hits, dist, _, face = RC.raycast_brute(soup, s, d)
        t64, f64 = _moller_trumbore_f64(v, f, s, d)
        assert np.array_equal(face >= 0, f64 >= 0)
        h = face >= 0
        assert h.sum() > 300
        np.testing.assert_allclose(dist[h], t64[h], rtol=1e-5, atol=1e-5)
        # the same triangle wins except where two candidates are closer than fp32 resolution (shared edges)
        same = face[h] == f64[h]
        assert same.mean() > 0.99
        exp = s[h].astype(np.float64) + t64[h, None] * d[h].astype(np.float64)
        np.testing.assert_allclose(hits[h], exp, rtol=1e-5, atol=2e-5)


# ---- GPU ------------------------------------------------------------------------------------------------
def _check_all_accels(device, v, f, s, d, max_dist=1e6):
    from isaaclab_amd.utils.warp import convert_to_warp_mesh, raycast_mesh

    soup = RC.tri_soup(v, f)
    exp = RC.raycast_brute(soup, s, d, max_dist)
    for accel in ("grid", "bvh"):
        mesh = convert_to_warp_mesh(v, f, str(device), accel=accel)
        assert mesh.info()["accel"] == accel
        for mode in (None, "brute"):
            got = raycast_mesh(to_dev(s, device), to_dev(d, device), mesh, max_dist, True, True, True, accel=mode)
            for name, g_, e_ in zip(("hits", "dist", "normal", "face"), got, exp):
                np.testing.assert_array_equal(to_np(g_), e_, err_msg=f"{name} accel={accel} mode={mode}")
    return exp


@pytest.mark.gpu
def test_gpu_plane_and_misses(device):
    v, f = plane(0.5, 10.0)
    s, d = rays_down(3, 5000, 14.0)   # ~half of the rays miss the 20 m plane
    exp = _check_all_accels(device, v, f, s, d)
    assert (exp[3] == -1).sum() > 500 and (exp[3] >= 0).sum() > 500
    _check_all_accels(device, v, f, s, d, max_dist=5.0)  # everything misses


@pytest.mark.gpu
@pytest.mark.parametrize("tilt", (0.0, 0.35))
def test_gpu_heightfield_bit_exact(device, tilt):
    v, f = heightfield(4, n=64)
    s, d = rays_down(5, 20000, 9.0, tilt=tilt)
    exp = _check_all_accels(device, v, f, s, d)
    assert (exp[3] >= 0).mean() > 0.3   # tilted rays partly leave the 16 m patch


@pytest.mark.gpu
def test_gpu_edge_and_vertex_ties(device):
    v, f = heightfield(6, n=24)
    s = v.copy()
    s[:, 2] = 10.0   # straight above every vertex: up to 6 faces share the hit
    d = np.tile(np.array([0, 0, -1.0], np.float32), (s.shape[0], 1))
    _check_all_accels(device, v, f, s, d)


@pytest.mark.gpu
def test_gpu_terrain_mesh_vs_oracle_grid(device):
    """The benchmark terrain generator (small variant): 187-ray grids at 256 poses, HIP vs the oracle grid caster."""
    from isaaclab_amd.terrains import SyntheticTerrain, SyntheticTerrainCfg
    from isaaclab_amd.terrains.synthetic import TerrainGeneratorCfg
    from isaaclab_amd.utils.warp import convert_to_warp_mesh, raycast_mesh
    from oracle import mdp_np as P

    terrain = SyntheticTerrain(SyntheticTerrainCfg(terrain_generator=TerrainGeneratorCfg(num_rows=3, num_cols=10)), 256)
    starts, dirs = P.grid_pattern(0.1, [1.6, 1.0])
    rng = np.random.default_rng(7)
    pos = terrain.env_origins + rng.normal(0, 0.5, (256, 3)).astype(np.float32)
    yaw = rng.uniform(-np.pi, np.pi, 256)
    quat = np.stack([np.cos(yaw / 2), 0 * yaw, 0 * yaw, np.sin(yaw / 2)], 1).astype(np.float32)
    sw, dw = P.raycaster_rays(pos, quat, starts + np.array([0, 0, 20.0], np.float32), dirs, True)
    sw, dw = sw.reshape(-1, 3), dw.reshape(-1, 3)
    exp = RC.GridCaster(RC.tri_soup(terrain.vertices, terrain.triangles)).cast(sw, dw)
    mesh = convert_to_warp_mesh(terrain.vertices, terrain.triangles, str(device))
    got = raycast_mesh(to_dev(sw, device), to_dev(dw, device), mesh, 1e6, True, True, True)
    for name, g_, e_ in zip(("hits", "dist", "normal", "face"), got, exp):
        np.testing.assert_array_equal(to_np(g_), e_, err_msg=name)
    assert (exp[3] >= 0).all()


@pytest.mark.gpu
def test_gpu_raycast_shapes_and_errors(device):
    import torch

    from isaaclab_amd.utils.warp import convert_to_warp_mesh, raycast_mesh

    v, f = plane(0.0, 5.0)
    mesh = convert_to_warp_mesh(v, f, str(device))
    s = torch.zeros(4, 7, 3, device=device)
    s[..., 2] = 3.0
    d = torch.zeros(4, 7, 3, device=device)
    d[..., 2] = -1.0
    hits, dist, normal, face = raycast_mesh(s, d, mesh, return_distance=True, return_face_id=True)
    assert hits.shape == (4, 7, 3) and dist.shape == (4, 7) and normal is None and face.shape == (4, 7)
    assert torch.allclose(dist, torch.full_like(dist, 3.0))
    with pytest.raises(ValueError):
        raycast_mesh(s, d[:, :3], mesh)


@pytest.mark.gpu
def test_gpu_vertical_ray_kernel_equals_generic_cast_bit_for_bit(device):
    """mdp_raycaster_update's vertical-ray routine (one cell per ray, zero-product terms dropped, walk ended by the
    disjoint-footprint rule) against the generic closest-hit cast on the SAME rays: the ray origins are taken from the
    routine's own output (a vertical ray keeps its xy; its start height is the sensor height + 20 m), so every hit
    height must be identical bit for bit -- on height-field tiles (disjoint cells) and on stair / box tiles (stacked
    faces, vertical walls) alike."""
    import torch

    from isaaclab_amd import _lib as L
    from isaaclab_amd.terrains import SyntheticTerrain, SyntheticTerrainCfg
    from isaaclab_amd.terrains.synthetic import TerrainGeneratorCfg
    from isaaclab_amd.utils.warp import convert_to_warp_mesh, raycast_mesh
    from oracle import mdp_np as P

    n = 1024
    terrain = SyntheticTerrain(SyntheticTerrainCfg(terrain_generator=TerrainGeneratorCfg(num_rows=3, num_cols=10)), n)
    mesh = convert_to_warp_mesh(terrain.vertices, terrain.triangles, str(device), accel="grid")
    assert mesh.info()["accel"] == "grid"
    starts, dirs = P.grid_pattern(0.1, [1.6, 1.0])
    starts = starts + np.array([0, 0, 20.0], np.float32)
    R = starts.shape[0]
    rng = np.random.default_rng(11)
    pos = (terrain.env_origins + rng.normal(0, 1.5, (n, 3)) * np.array([1, 1, 0.1])).astype(np.float32)
    yaw = rng.uniform(-np.pi, np.pi, n)
    yaw[:64] = 0.0
    pose = np.concatenate([pos, np.stack([np.cos(yaw / 2), 0 * yaw, 0 * yaw, np.sin(yaw / 2)], 1)], 1).astype(np.float32)
    d_pose, d_starts, d_dirs = to_dev(pose, device), to_dev(starts, device), to_dev(dirs, device)
    pos_w, quat_w = torch.zeros(n, 3, device=device), torch.zeros(n, 4, device=device)
    hits = torch.zeros(n, R, 3, device=device)
    L.check(L.lib.mdp_raycaster_update(mesh.handle, n, R, d_pose.data_ptr(), 7, None, d_starts.data_ptr(),
                                       d_dirs.data_ptr(), 1, 1, 1e6, pos_w.data_ptr(), quat_w.data_ptr(),
                                       hits.data_ptr(), None, None, None, None, 0, 0.0, 0.0, L.stream(device)))
    torch.cuda.synchronize()
    h = to_np(hits)
    hit = np.isfinite(h[..., 2])                     # a few rays of envs near the border leave the terrain
    assert hit.mean() > 0.98 and np.all(np.isinf(h[~hit]))
    # the same rays through the generic path
    s = h.copy()
    s[..., 2] = (np.float32(20.0) + pos[:, None, 2]).astype(np.float32)
    s[~hit] = np.array([0.0, 0.0, 1e9], np.float32)  # placeholders, not compared
    d = np.tile(np.array([0, 0, -1.0], np.float32), (n, R, 1))
    ref = to_np(raycast_mesh(to_dev(s, device), to_dev(d, device), mesh, 1e6)[0])
    np.testing.assert_array_equal(h[hit], ref[hit])
    # yaw == 0 envs: the ray origins themselves are exact sums, so the oracle's grid caster must agree as well
    s0 = (starts[None, :, :] + pos[:64, None, :]).astype(np.float32).reshape(-1, 3)
    exp = RC.GridCaster(RC.tri_soup(terrain.vertices, terrain.triangles)).cast(s0, d[:64].reshape(-1, 3))
    np.testing.assert_array_equal(h[:64].reshape(-1, 3), exp[0])
