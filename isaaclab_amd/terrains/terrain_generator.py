"""Terrain generator and env-origin layout without trimesh / Isaac Sim.

``TerrainGenerator`` restates ``LAB/terrains/terrain_generator.py:96-387``: sub-terrain selection by proportion per
column (curriculum, ``:223-254``) or by ``np_rng.choice`` (random, ``:202-221``), difficulty from
``np.random.default_rng(seed)``, each sub-terrain centred, moved to its tile, the border frame with only its top faces
kept (``:260-280``), everything concatenated and centred on the origin (``:169-175``).  ``ROUGH_TERRAINS_CFG`` is
``LAB/terrains/config/rough.py:12-51``.  ``GeneratedTerrain`` is the numpy / torch level of ``TerrainImporter``
(``LAB/terrains/terrain_importer.py:284-349``): mesh as fp32 vertices + int32 triangles (what ``convert_to_warp_mesh``
receives), tile origins, ``terrain_levels`` / ``terrain_types`` / ``env_origins`` of the curriculum layout.

Not here (not on the hot path): the on-disk cache, colour schemes, flat-patch sampling (``find_flat_patches`` casts rays
through warp at construction time), USD import.  Host-side input generation: no kernels, not a performance target."""
from __future__ import annotations

from dataclasses import MISSING

import numpy as np
import torch

from ..utils.configclass import configclass
from . import height_field as hf
from . import mesh_terrains as mt
from .mesh_terrains import Mesh, concatenate, make_border


@configclass
class TerrainGeneratorCfg:
    """terrain_generator_cfg.py:100-200."""

    seed: int | None = None
    curriculum: bool = False
    size: tuple = MISSING
    border_width: float = 0.0
    border_height: float = 1.0
    num_rows: int = 1
    num_cols: int = 1
    color_scheme: str = "none"
    horizontal_scale: float = 0.1
    vertical_scale: float = 0.005
    slope_threshold: float | None = 0.75
    sub_terrains: dict = MISSING
    difficulty_range: tuple = (0.0, 1.0)
    use_cache: bool = False
    cache_dir: str = "/tmp/isaaclab/terrains"


ROUGH_TERRAINS_CFG = TerrainGeneratorCfg(
    size=(8.0, 8.0),
    border_width=20.0,
    num_rows=10,
    num_cols=20,
    horizontal_scale=0.1,
    vertical_scale=0.005,
    slope_threshold=0.75,
    use_cache=False,
    sub_terrains={
        "pyramid_stairs": mt.MeshPyramidStairsTerrainCfg(
            proportion=0.2, step_height_range=(0.05, 0.23), step_width=0.3, platform_width=3.0, border_width=1.0, holes=False),
        "pyramid_stairs_inv": mt.MeshInvertedPyramidStairsTerrainCfg(
            proportion=0.2, step_height_range=(0.05, 0.23), step_width=0.3, platform_width=3.0, border_width=1.0, holes=False),
        "boxes": mt.MeshRandomGridTerrainCfg(proportion=0.2, grid_width=0.45, grid_height_range=(0.05, 0.2), platform_width=2.0),
        "random_rough": hf.HfRandomUniformTerrainCfg(proportion=0.2, noise_range=(0.02, 0.10), noise_step=0.02, border_width=0.25),
        "hf_pyramid_slope": hf.HfPyramidSlopedTerrainCfg(proportion=0.1, slope_range=(0.0, 0.4), platform_width=2.0, border_width=0.25),
        "hf_pyramid_slope_inv": hf.HfInvertedPyramidSlopedTerrainCfg(proportion=0.1, slope_range=(0.0, 0.4), platform_width=2.0, border_width=0.25),
    },
)
"""Rough terrains configuration (config/rough.py:12-51)."""


class TerrainGenerator:
    """terrain_generator.py:24-387.  ``terrain_mesh`` is one ``Mesh`` (float64 vertices, like trimesh) for all
    sub-terrains plus the border; ``terrain_origins`` is (num_rows, num_cols, 3)."""

    def __init__(self, cfg: TerrainGeneratorCfg, device: str = "cpu"):
        if len(cfg.sub_terrains) == 0:
            raise ValueError("No sub-terrains specified! Please add at least one sub-terrain.")
        self.cfg = cfg
        self.device = device
        for sub_cfg in self.cfg.sub_terrains.values():
            sub_cfg.size = self.cfg.size
            if isinstance(sub_cfg, hf.HfTerrainBaseCfg):
                sub_cfg.horizontal_scale = self.cfg.horizontal_scale
                sub_cfg.vertical_scale = self.cfg.vertical_scale
                sub_cfg.slope_threshold = self.cfg.slope_threshold
        if self.cfg.use_cache:
            raise NotImplementedError("the on-disk sub-terrain cache (terrain_generator.py:353-385) is not part of this build")
        seed = self.cfg.seed if self.cfg.seed is not None else np.random.get_state()[1][0]
        self.np_rng = np.random.default_rng(seed)
        self.flat_patches: dict = {}
        self.terrain_meshes: list[Mesh] = []
        self.terrain_origins = np.zeros((self.cfg.num_rows, self.cfg.num_cols, 3))
        if self.cfg.curriculum:
            self._generate_curriculum_terrains()
        else:
            self._generate_random_terrains()
        self._add_terrain_border()
        self.terrain_mesh = concatenate(self.terrain_meshes)
        if self.cfg.color_scheme not in ("height", "random", "none"):
            raise ValueError(f"Invalid color scheme: {self.cfg.color_scheme}.")
        if self.cfg.color_scheme == "random":     # the draw is part of the random stream; colours themselves are not kept
            self.np_rng.choice(range(256), size=(len(self.terrain_mesh.vertices), 4))
        transform = np.eye(4)
        transform[:2, -1] = -self.cfg.size[0] * self.cfg.num_rows * 0.5, -self.cfg.size[1] * self.cfg.num_cols * 0.5
        self.terrain_mesh.apply_transform(transform)
        self.terrain_origins += transform[:3, -1]

    def __str__(self):
        msg = "Terrain Generator:"
        msg += f"\n\tSeed: {self.cfg.seed}"
        msg += f"\n\tNumber of rows: {self.cfg.num_rows}"
        msg += f"\n\tNumber of columns: {self.cfg.num_cols}"
        msg += f"\n\tSub-terrain size: {self.cfg.size}"
        msg += f"\n\tSub-terrain types: {list(self.cfg.sub_terrains.keys())}"
        msg += f"\n\tCurriculum: {self.cfg.curriculum}"
        msg += f"\n\tDifficulty range: {self.cfg.difficulty_range}"
        msg += f"\n\tColor scheme: {self.cfg.color_scheme}"
        msg += f"\n\tUse cache: {self.cfg.use_cache}"
        return msg

    def _proportions(self) -> np.ndarray:
        p = np.array([sub_cfg.proportion for sub_cfg in self.cfg.sub_terrains.values()])
        return p / np.sum(p)

    def _generate_random_terrains(self):
        proportions = self._proportions()
        cfgs = list(self.cfg.sub_terrains.values())
        for index in range(self.cfg.num_rows * self.cfg.num_cols):
            row, col = np.unravel_index(index, (self.cfg.num_rows, self.cfg.num_cols))
            sub_index = self.np_rng.choice(len(proportions), p=proportions)
            difficulty = self.np_rng.uniform(*self.cfg.difficulty_range)
            mesh, origin = self._get_terrain_mesh(difficulty, cfgs[sub_index])
            self._add_sub_terrain(mesh, origin, int(row), int(col), cfgs[sub_index])

    def _generate_curriculum_terrains(self):
        cum = np.cumsum(self._proportions())
        sub_indices = [int(np.min(np.where(index / self.cfg.num_cols + 0.001 < cum)[0])) for index in range(self.cfg.num_cols)]
        cfgs = list(self.cfg.sub_terrains.values())
        for col in range(self.cfg.num_cols):
            for row in range(self.cfg.num_rows):
                lower, upper = self.cfg.difficulty_range
                difficulty = (row + self.np_rng.uniform()) / self.cfg.num_rows
                difficulty = lower + (upper - lower) * difficulty
                mesh, origin = self._get_terrain_mesh(difficulty, cfgs[sub_indices[col]])
                self._add_sub_terrain(mesh, origin, row, col, cfgs[sub_indices[col]])

    def _add_terrain_border(self):
        border_size = (self.cfg.num_rows * self.cfg.size[0] + 2 * self.cfg.border_width,
                       self.cfg.num_cols * self.cfg.size[1] + 2 * self.cfg.border_width)
        inner_size = (self.cfg.num_rows * self.cfg.size[0], self.cfg.num_cols * self.cfg.size[1])
        center = (self.cfg.num_rows * self.cfg.size[0] / 2, self.cfg.num_cols * self.cfg.size[1] / 2, -self.cfg.border_height / 2)
        border = concatenate(make_border(border_size, inner_size, height=self.cfg.border_height, position=center))
        # only the faces without a vertex below z = -0.1, i.e. the top of the frame
        border.update_faces(~(border.triangles[:, :, 2] < -0.1).any(1))
        self.terrain_meshes.append(border)

    def _add_sub_terrain(self, mesh: Mesh, origin: np.ndarray, row: int, col: int, sub_terrain_cfg):
        if sub_terrain_cfg.flat_patch_sampling is not None:
            raise NotImplementedError("flat-patch sampling (terrains/utils.py:84-275) is not part of this build")
        transform = np.eye(4)
        transform[0:2, -1] = (row + 0.5) * self.cfg.size[0], (col + 0.5) * self.cfg.size[1]
        mesh.apply_transform(transform)
        self.terrain_meshes.append(mesh)
        self.terrain_origins[row, col] = origin + transform[:3, -1]

    def _get_terrain_mesh(self, difficulty: float, cfg) -> tuple[Mesh, np.ndarray]:
        cfg = cfg.copy()
        cfg.difficulty = float(difficulty)
        cfg.seed = self.cfg.seed
        meshes, origin = cfg.function(difficulty, cfg)
        mesh = concatenate(meshes)
        transform = np.eye(4)
        transform[0:2, -1] = -cfg.size[0] * 0.5, -cfg.size[1] * 0.5
        mesh.apply_transform(transform)
        origin = origin + transform[0:3, -1]
        return mesh, origin


@configclass
class TerrainImporterCfg:
    """The fields of terrain_importer_cfg.py:28-103 the layout reads."""

    terrain_type: str = "generator"
    terrain_generator: TerrainGeneratorCfg | None = None
    max_init_terrain_level: int | None = None
    num_envs: int = 1
    env_spacing: float | None = None


class GeneratedTerrain:
    """``TerrainImporter`` for ``terrain_type='generator'`` without the simulator: the generated mesh and the curriculum
    layout of the envs (terrain_importer.py:284-349).  The initial levels come from ``torch.randint`` on the global CPU
    generator, like the reference's; ``terrain_types`` is ``floor(arange(N) / (N / num_cols))``."""

    def __init__(self, cfg, num_envs: int):
        if cfg.terrain_generator is None:
            raise ValueError("Input terrain type is 'generator' but no value provided for 'terrain_generator'.")
        self.cfg = cfg
        gen = TerrainGenerator(cfg.terrain_generator)
        self.generator = gen
        # what convert_to_warp_mesh receives (ops.py:130-145): fp32 points, int32 indices
        self.vertices = np.ascontiguousarray(gen.terrain_mesh.vertices, dtype=np.float32)
        self.triangles = np.ascontiguousarray(gen.terrain_mesh.faces, dtype=np.int32)
        origins = torch.from_numpy(gen.terrain_origins).to(dtype=torch.float)
        self.terrain_origins = origins.numpy()
        num_rows, num_cols = origins.shape[:2]
        max_init = num_rows - 1 if cfg.max_init_terrain_level is None else min(cfg.max_init_terrain_level, num_rows - 1)
        self._max_terrain_level = num_rows
        levels = torch.randint(0, max_init + 1, (num_envs,))
        types_ = torch.div(torch.arange(num_envs), (num_envs / num_cols), rounding_mode="floor").to(torch.long)
        self.terrain_levels = levels.numpy()
        self.terrain_types = types_.numpy()
        self.env_origins = origins[levels, types_].numpy().astype(np.float32)

    @property
    def max_terrain_level(self) -> int:
        return self._max_terrain_level

    def device_state(self, device):
        """Device copies of what the terrain curriculum reads and writes (terrain_importer.py:307-324)."""
        st = self.__dict__.get("_device_state")
        if st is None or st["device"] != str(device):
            st = dict(device=str(device),
                      terrain_levels=torch.as_tensor(np.asarray(self.terrain_levels, np.int64), device=device).clone(),
                      terrain_types=torch.as_tensor(np.asarray(self.terrain_types, np.int64), device=device).clone(),
                      terrain_origins=torch.as_tensor(self.terrain_origins, device=device).contiguous().clone())
            self._device_state = st
        return st
