"""Mesh sub-terrains, TerrainGenerator and the env-origin layout against vectors produced by the REFERENCE code
(tests/golden/make_golden.py: gen_mesh_terrains ran trimesh/mesh_terrains.py, trimesh/utils.py, terrain_generator.py and
TerrainImporter._compute_env_origins_curriculum; trimesh's box primitive came from the stand-in documented in
tests/golden/trimesh_standin.py -- its vertex / face order is "parity unpinned", everything else is pinned).
The comparison is the triangle soup in face order, fp32: what the ray caster receives."""
import hashlib
import os

import numpy as np
import pytest
import torch

from isaaclab_amd.terrains import mesh_terrains as MT
from isaaclab_amd.terrains import terrain_generator as TG

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "mesh_terrains.npz"))


def soup(mesh):
    return np.asarray(mesh.vertices, np.float64)[np.asarray(mesh.faces)].astype(np.float32)


SUBS = {
    "stairs": lambda: MT.MeshPyramidStairsTerrainCfg(size=(8.0, 8.0), step_height_range=(0.05, 0.23), step_width=0.3,
                                                     platform_width=3.0, border_width=1.0, holes=False),
    "stairs_inv": lambda: MT.MeshInvertedPyramidStairsTerrainCfg(size=(8.0, 8.0), step_height_range=(0.05, 0.23), step_width=0.3,
                                                                 platform_width=3.0, border_width=1.0, holes=False),
    "stairs_holes": lambda: MT.MeshPyramidStairsTerrainCfg(size=(6.0, 7.0), step_height_range=(0.1, 0.2), step_width=0.25,
                                                           platform_width=1.5, border_width=0.0, holes=True),
    "boxes": lambda: MT.MeshRandomGridTerrainCfg(size=(8.0, 8.0), grid_width=0.45, grid_height_range=(0.05, 0.2), platform_width=2.0),
    "boxes_holes": lambda: MT.MeshRandomGridTerrainCfg(size=(6.0, 6.0), grid_width=0.35, grid_height_range=(0.02, 0.1),
                                                       platform_width=1.5, holes=True),
    "flat": lambda: MT.MeshPlaneTerrainCfg(size=(5.0, 4.0)),
}


@pytest.mark.parametrize("k,name", list(enumerate(SUBS)))
def test_mesh_sub_terrains_equal_the_reference(k, name):
    cfg = SUBS[name]()
    torch.manual_seed(300 + k)          # random_grid_terrain draws from torch's global generator, like the reference
    meshes, origin = cfg.function(float(G[name + "_difficulty"][0]), cfg)
    np.testing.assert_array_equal(soup(MT.concatenate(meshes)), G[name + "_soup"])
    np.testing.assert_array_equal(np.asarray(origin, np.float64), G[name + "_origin"])


def rough(rows, cols, curriculum, seed):
    cfg = TG.ROUGH_TERRAINS_CFG.copy()
    cfg.num_rows, cfg.num_cols, cfg.curriculum, cfg.seed = rows, cols, curriculum, seed
    return cfg


@pytest.mark.parametrize("name", ["gen_cur", "gen_rand"])
def test_generator_layouts_equal_the_reference(name):
    rows, cols, cur, seed = (int(v) for v in G[name + "_layout"])
    np.random.seed(seed + 1)            # the height-field noise uses numpy's global generator
    torch.manual_seed(seed + 2)
    g = TG.TerrainGenerator(rough(rows, cols, bool(cur), seed))
    np.testing.assert_array_equal(soup(g.terrain_mesh), G[name + "_soup"])
    np.testing.assert_array_equal(g.terrain_origins, G[name + "_origins"])


def test_rough_terrains_cfg_full_size():
    """ROUGH_TERRAINS_CFG with the curriculum layout (velocity_env_cfg.py:314-318), seed 42: digest of the whole fp32
    soup, samples, tile origins and the 4096-env origin layout."""
    np.random.seed(42)
    torch.manual_seed(42)
    cfg = rough(10, 20, True, 42)
    torch.manual_seed(42)
    np.random.seed(42)
    gen = TG.TerrainGenerator(cfg)
    sp = soup(gen.terrain_mesh)
    assert sp.shape[0] == int(G["rough_num_tris"][0])
    np.testing.assert_array_equal(sp[::4099], G["rough_sample"])
    digest = np.frombuffer(hashlib.sha256(np.ascontiguousarray(sp).tobytes()).digest(), dtype=np.uint8)
    np.testing.assert_array_equal(digest, G["rough_digest"])
    np.testing.assert_array_equal(gen.terrain_origins, G["rough_origins"])
    # GeneratedTerrain = generator + TerrainImporter layout (same seeds -> same mesh; levels drawn after it)
    np.random.seed(42)
    torch.manual_seed(42)
    real_init = TG.TerrainGenerator.__init__

    def seeded_init(self, *a, **k):
        real_init(self, *a, **k)
        torch.manual_seed(43)            # make_golden seeds the layout draw separately

    TG.TerrainGenerator.__init__ = seeded_init
    try:
        t = TG.GeneratedTerrain(TG.TerrainImporterCfg(terrain_generator=cfg, max_init_terrain_level=5, num_envs=4096), 4096)
    finally:
        TG.TerrainGenerator.__init__ = real_init
    np.testing.assert_array_equal(t.terrain_levels, G["rough_levels"])
    np.testing.assert_array_equal(t.terrain_types, G["rough_types"])
    np.testing.assert_array_equal(t.env_origins, G["rough_env_origins"])
    assert t.vertices.dtype == np.float32 and t.triangles.dtype == np.int32 and t.max_terrain_level == 10


def test_generator_errors():
    with pytest.raises(ValueError):
        TG.TerrainGenerator(TG.TerrainGeneratorCfg(size=(8.0, 8.0), sub_terrains={}))
    with pytest.raises(ValueError):
        MT.random_grid_terrain(0.5, MT.MeshRandomGridTerrainCfg(size=(8.0, 6.0), grid_width=0.45, grid_height_range=(0.05, 0.2)))
    with pytest.raises(RuntimeError):
        MT.random_grid_terrain(0.5, MT.MeshRandomGridTerrainCfg(size=(6.0, 6.0), grid_width=0.4, grid_height_range=(0.05, 0.2)))
