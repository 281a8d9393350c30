"""Height-field sub-terrains (first slice) against golden vectors the reference produced
(tests/golden/make_golden.py::gen_terrains: hf_terrains.py + height_field/utils.py with trimesh.Trimesh replaced by a
recorder, np.random seeded).  Host-side numpy: vertices, faces and origins must be identical."""
import os

import numpy as np

from isaaclab_amd.terrains import height_field as HF

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "terrains.npz"))
BASE = dict(size=(6.0, 5.0), horizontal_scale=0.1, vertical_scale=0.005, slope_threshold=0.75)
CASES = {
    "random": (HF.random_uniform_terrain, HF.HfRandomUniformTerrainCfg,
               dict(border_width=0.25, noise_range=(0.02, 0.10), noise_step=0.02, downsampled_scale=None)),
    "random_ds": (HF.random_uniform_terrain, HF.HfRandomUniformTerrainCfg,
                  dict(border_width=0.0, noise_range=(-0.05, 0.05), noise_step=0.01, downsampled_scale=0.2)),
    "slope": (HF.pyramid_sloped_terrain, HF.HfPyramidSlopedTerrainCfg,
              dict(border_width=0.25, slope_range=(0.0, 0.4), platform_width=2.0, inverted=False)),
    "slope_inv": (HF.pyramid_sloped_terrain, HF.HfPyramidSlopedTerrainCfg,
                  dict(border_width=0.25, slope_range=(0.0, 0.4), platform_width=2.0, inverted=True)),
    "stairs": (HF.pyramid_stairs_terrain, HF.HfPyramidStairsTerrainCfg,
               dict(border_width=0.5, step_height_range=(0.05, 0.23), step_width=0.3, platform_width=1.5, inverted=False)),
    "stairs_inv": (HF.pyramid_stairs_terrain, HF.HfPyramidStairsTerrainCfg,
                   dict(border_width=0.5, step_height_range=(0.05, 0.23), step_width=0.3, platform_width=1.5, inverted=True)),
}


def test_height_field_terrains_equal_the_reference():
    for k, (name, (fn, cfg_cls, extra)) in enumerate(CASES.items()):
        cfg = cfg_cls(**BASE, **extra)
        np.random.seed(100 + k)
        meshes, origin = fn(float(G[name + "_difficulty"][0]), cfg)
        assert tuple(cfg.size) == (6.0, 5.0)                       # the wrapper restores the tile size
        np.testing.assert_array_equal(meshes[0].vertices, G[name + "_vertices"], err_msg=name)
        np.testing.assert_array_equal(meshes[0].faces, G[name + "_faces"], err_msg=name)
        np.testing.assert_array_equal(origin, G[name + "_origin"], err_msg=name)
        z = meshes[0].vertices[:, 2]
        assert z.max() > 0.04 or z.min() < -0.04, name              # not a flat tile


def test_border_width_must_cover_a_cell():
    import pytest

    cfg = HF.HfPyramidSlopedTerrainCfg(**BASE, border_width=0.05)
    with pytest.raises(ValueError, match="border width"):
        HF.pyramid_sloped_terrain(0.5, cfg)
