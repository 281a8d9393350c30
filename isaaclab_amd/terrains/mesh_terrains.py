"""Box-built sub-terrains without trimesh: ``LAB/terrains/trimesh/utils.py:15-97``
(``make_plane``, ``make_border``) and ``LAB/terrains/trimesh/mesh_terrains.py:23-376`` (``flat_terrain``,
``pyramid_stairs_terrain``, ``inverted_pyramid_stairs_terrain``, ``random_grid_terrain``) -- the mesh sub-terrains of
``ROUGH_TERRAINS_CFG`` (``LAB/terrains/config/rough.py:12-51``).  Same names, ``(difficulty, cfg)`` arguments, box
dimensions / positions / order and random stream (the grid heights come from ``Tensor.uniform_`` on torch's global CPU
generator) as the reference; a sub-terrain comes back as ``([Mesh, ...], origin)`` where the reference returns
``trimesh.Trimesh`` objects.

The one thing the reference does not hold is the box primitive itself: ``trimesh.creation.box`` lives in the absent
``trimesh`` package.  ``box()`` below restates its published layout (eight corners in binary (x, y, z) order, twelve
hard-coded triangles); vertex NUMBERING after trimesh's duplicate-vertex merge is not reproduced (nothing on the hot
path reads it) -- the triangle soup in face order is what the ray caster consumes and what the golden vectors pin.
Host-side input generation in float64 like trimesh, cast to fp32 where the reference hands the mesh to the ray caster
(``LAB/utils/warp/ops.py:130-145``): no kernels, not a performance target."""
from __future__ import annotations

from dataclasses import MISSING

import numpy as np
import torch

from ..utils.configclass import configclass


class Mesh:
    """Vertices (V, 3) float64 + faces (F, 3) int64: the slice of ``trimesh.Trimesh`` the terrain code uses."""

    def __init__(self, vertices, faces):
        self.vertices = np.array(vertices, dtype=np.float64).reshape(-1, 3)
        self.faces = np.array(faces, dtype=np.int64).reshape(-1, 3)

    @property
    def triangles(self) -> np.ndarray:
        return self.vertices[self.faces]

    def apply_translation(self, t) -> "Mesh":
        self.vertices = self.vertices + np.asarray(t, dtype=np.float64)
        return self

    def apply_transform(self, matrix) -> "Mesh":
        m = np.asarray(matrix, dtype=np.float64)
        if not np.array_equal(m[:3, :3], np.eye(3)):
            self.vertices = self.vertices @ m[:3, :3].T
        return self.apply_translation(m[:3, 3])

    def update_faces(self, mask) -> None:
        self.faces = self.faces[np.asarray(mask)]

    def copy(self) -> "Mesh":
        return Mesh(self.vertices.copy(), self.faces.copy())


def concatenate(meshes) -> Mesh:
    """``trimesh.util.concatenate``: vertices appended, face indices shifted, order kept."""
    meshes = list(meshes)
    if len(meshes) == 0:
        return Mesh(np.zeros((0, 3)), np.zeros((0, 3), np.int64))
    offsets = np.cumsum([0] + [len(m.vertices) for m in meshes[:-1]])
    return Mesh(np.concatenate([np.asarray(m.vertices, np.float64).reshape(-1, 3) for m in meshes]),
                np.concatenate([np.asarray(m.faces, np.int64).reshape(-1, 3) + o for m, o in zip(meshes, offsets)]))


_BOX_CORNERS = np.array([[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [1, 0, 0], [1, 0, 1], [1, 1, 0], [1, 1, 1]],
                        dtype=np.float64) - 0.5
_BOX_FACES = np.array([1, 3, 0, 4, 1, 0, 0, 3, 2, 2, 4, 0, 1, 7, 3, 5, 1, 4, 5, 7, 1, 3, 7, 2, 6, 4, 2, 2, 7, 6, 6, 5, 4,
                       7, 5, 6], dtype=np.int64).reshape(-1, 3)


def box(extents, position=(0.0, 0.0, 0.0)) -> Mesh:
    """``trimesh.creation.box(extents, translation_matrix(position))``: unit cube centred at the origin, scaled by the
    extents, then moved."""
    v = _BOX_CORNERS * np.asarray(extents, dtype=np.float64)
    return Mesh(v + np.asarray(position, dtype=np.float64), _BOX_FACES.copy())


def make_plane(size, height: float, center_zero: bool = True) -> Mesh:
    """utils.py:15-45: two triangles, (0, 0) .. size or centred."""
    v = np.array([[size[0], size[1], height], [size[0], 0.0, height], [0.0, size[1], height], [0.0, 0.0, height]])
    m = Mesh(v, [[1, 0, 2], [2, 3, 1]])
    if center_zero:
        m.apply_translation(-np.array([size[0] / 2.0, size[1] / 2.0, 0.0]))
    return m


def make_border(size, inner_size, height: float, position) -> list[Mesh]:
    """utils.py:48-97: four boxes around a rectangular hole, returned as [left, right, top, bottom]."""
    tx = (size[0] - inner_size[0]) / 2.0
    ty = (size[1] - inner_size[1]) / 2.0
    dims = (size[0], ty, height)
    top = box(dims, (position[0], position[1] + inner_size[1] / 2.0 + ty / 2.0, position[2]))
    bottom = box(dims, (position[0], position[1] - inner_size[1] / 2.0 - ty / 2.0, position[2]))
    dims = (tx, inner_size[1], height)
    left = box(dims, (position[0] - inner_size[0] / 2.0 - tx / 2.0, position[1], position[2]))
    right = box(dims, (position[0] + inner_size[0] / 2.0 + tx / 2.0, position[1], position[2]))
    return [left, right, top, bottom]


# ---- configs (mesh_terrains_cfg.py:20-82, terrain_generator_cfg.py:60-97) ----------------------------------------------


@configclass
class SubTerrainBaseCfg:
    function: object = MISSING
    proportion: float = 1.0
    size: tuple = (10.0, 10.0)
    flat_patch_sampling: dict | None = None


def flat_terrain(difficulty: float, cfg) -> tuple[list[Mesh], np.ndarray]:
    """mesh_terrains.py:23-47."""
    origin = (cfg.size[0] / 2.0, cfg.size[1] / 2.0, 0.0)
    return [make_plane(cfg.size, 0.0, center_zero=False)], np.array(origin)


def _stair_rings(cfg, step_height: float, inverted: bool) -> tuple[list[Mesh], np.ndarray]:
    """mesh_terrains.py:74-147 / 174-247: border frame, one ring of four boxes per step, the middle box."""
    nx = (cfg.size[0] - 2 * cfg.border_width - cfg.platform_width) // (2 * cfg.step_width) + 1
    ny = (cfg.size[1] - 2 * cfg.border_width - cfg.platform_width) // (2 * cfg.step_width) + 1
    num_steps = int(min(nx, ny))
    total_height = (num_steps + 1) * step_height
    meshes: list[Mesh] = []
    if cfg.border_width > 0.0 and not cfg.holes:
        center = [0.5 * cfg.size[0], 0.5 * cfg.size[1], -0.5 * step_height if inverted else -step_height / 2]
        inner = (cfg.size[0] - 2 * cfg.border_width, cfg.size[1] - 2 * cfg.border_width)
        meshes += make_border(cfg.size, inner, step_height, center)
    c = [0.5 * cfg.size[0], 0.5 * cfg.size[1], 0.0]
    ts = (cfg.size[0] - 2 * cfg.border_width, cfg.size[1] - 2 * cfg.border_width)
    for k in range(num_steps):
        bs = (cfg.platform_width, cfg.platform_width) if cfg.holes else (ts[0] - 2 * k * cfg.step_width, ts[1] - 2 * k * cfg.step_width)
        if inverted:
            z = c[2] - total_height / 2 - (k + 1) * step_height / 2.0
            h = total_height - (k + 1) * step_height
        else:
            z = c[2] + k * step_height / 2.0
            h = (k + 2) * step_height
        off = (k + 0.5) * cfg.step_width
        dims = (bs[0], cfg.step_width, h)
        top = box(dims, (c[0], c[1] + ts[1] / 2.0 - off, z))
        bottom = box(dims, (c[0], c[1] - ts[1] / 2.0 + off, z))
        dims = (cfg.step_width, bs[1], h) if cfg.holes else (cfg.step_width, bs[1] - 2 * cfg.step_width, h)
        right = box(dims, (c[0] + ts[0] / 2.0 - off, c[1], z))
        left = box(dims, (c[0] - ts[0] / 2.0 + off, c[1], z))
        meshes += [top, bottom, right, left]
    mid = (ts[0] - 2 * num_steps * cfg.step_width, ts[1] - 2 * num_steps * cfg.step_width)
    if inverted:
        meshes.append(box((mid[0], mid[1], step_height), (c[0], c[1], c[2] - total_height - step_height / 2)))
        origin = np.array([c[0], c[1], -(num_steps + 1) * step_height])
    else:
        meshes.append(box((mid[0], mid[1], (num_steps + 2) * step_height), (c[0], c[1], c[2] + num_steps * step_height / 2)))
        origin = np.array([c[0], c[1], (num_steps + 1) * step_height])
    return meshes, origin


def pyramid_stairs_terrain(difficulty: float, cfg) -> tuple[list[Mesh], np.ndarray]:
    """mesh_terrains.py:50-147."""
    step_height = cfg.step_height_range[0] + difficulty * (cfg.step_height_range[1] - cfg.step_height_range[0])
    return _stair_rings(cfg, step_height, inverted=False)


def inverted_pyramid_stairs_terrain(difficulty: float, cfg) -> tuple[list[Mesh], np.ndarray]:
    """mesh_terrains.py:150-247."""
    step_height = cfg.step_height_range[0] + difficulty * (cfg.step_height_range[1] - cfg.step_height_range[0])
    return _stair_rings(cfg, step_height, inverted=True)


def random_grid_terrain(difficulty: float, cfg) -> tuple[list[Mesh], np.ndarray]:
    """mesh_terrains.py:250-376: a grid of 1 m tall boxes whose tops are shifted by U(-h, h) (one draw per box from
    torch's global generator, like the reference's ``h_noise[:, 2].uniform_``), a border frame and the platform."""
    if cfg.size[0] != cfg.size[1]:
        raise ValueError(f"The terrain must be square. Received size: {cfg.size}.")
    grid_height = cfg.grid_height_range[0] + difficulty * (cfg.grid_height_range[1] - cfg.grid_height_range[0])
    meshes: list[Mesh] = []
    nbx, nby = int(cfg.size[0] / cfg.grid_width), int(cfg.size[1] / cfg.grid_width)
    terrain_height = 1.0
    border_width = cfg.size[0] - min(nbx, nby) * cfg.grid_width
    if border_width > 0:
        center = (0.5 * cfg.size[0], 0.5 * cfg.size[1], -terrain_height / 2)
        inner = (cfg.size[0] - border_width, cfg.size[1] - border_width)
        meshes += make_border(cfg.size, inner, terrain_height, center)
    else:
        raise RuntimeError("Border width must be greater than 0! Adjust the parameter 'cfg.grid_width'.")
    template = box([cfg.grid_width, cfg.grid_width, terrain_height], [0.5 * cfg.grid_width, 0.5 * cfg.grid_width, -terrain_height / 2])
    vertices = torch.tensor(template.vertices).repeat(nbx * nby, 1, 1)
    xx, yy = torch.meshgrid(torch.arange(0, nbx), torch.arange(0, nby), indexing="ij")
    xx_yy = torch.cat((xx.flatten().view(-1, 1), yy.flatten().view(-1, 1)), dim=1)
    offsets = cfg.grid_width * xx_yy + border_width / 2
    vertices[:, :, :2] += offsets.unsqueeze(1)
    if cfg.holes:
        mask_x = torch.logical_and(
            (vertices[:, :, 0] > (cfg.size[0] - border_width - cfg.platform_width) / 2).all(dim=1),
            (vertices[:, :, 0] < (cfg.size[0] + border_width + cfg.platform_width) / 2).all(dim=1))
        mask_y = torch.logical_and(
            (vertices[:, :, 1] > (cfg.size[1] - border_width - cfg.platform_width) / 2).all(dim=1),
            (vertices[:, :, 1] < (cfg.size[1] + border_width + cfg.platform_width) / 2).all(dim=1))
        vertices = torch.cat((vertices[mask_x], vertices[mask_y]))
    num_boxes = len(vertices)
    h_noise = torch.zeros((num_boxes, 3))
    h_noise[:, 2].uniform_(-grid_height, grid_height)
    vertices_noise = torch.zeros((num_boxes, 4, 3))
    vertices_noise += h_noise.unsqueeze(1)
    vertices[vertices[:, :, 2] == 0] += vertices_noise.view(-1, 3)
    faces = torch.tensor(template.faces).repeat(num_boxes, 1, 1)
    faces += (torch.arange(0, num_boxes).unsqueeze(1).repeat(1, 12) * 8).unsqueeze(2)
    meshes.append(Mesh(vertices.reshape(-1, 3).numpy(), faces.view(-1, 3).numpy()))
    dim = (cfg.platform_width, cfg.platform_width, terrain_height + grid_height)
    pos = (0.5 * cfg.size[0], 0.5 * cfg.size[1], -terrain_height / 2 + grid_height / 2)
    meshes.append(box(dim, pos))
    return meshes, np.array([0.5 * cfg.size[0], 0.5 * cfg.size[1], grid_height])


@configclass
class MeshPlaneTerrainCfg(SubTerrainBaseCfg):
    function = flat_terrain


@configclass
class MeshPyramidStairsTerrainCfg(SubTerrainBaseCfg):
    function = pyramid_stairs_terrain
    border_width: float = 0.0
    step_height_range: tuple = MISSING
    step_width: float = MISSING
    platform_width: float = 1.0
    holes: bool = False


@configclass
class MeshInvertedPyramidStairsTerrainCfg(MeshPyramidStairsTerrainCfg):
    function = inverted_pyramid_stairs_terrain


@configclass
class MeshRandomGridTerrainCfg(SubTerrainBaseCfg):
    function = random_grid_terrain
    grid_width: float = MISSING
    grid_height_range: tuple = MISSING
    platform_width: float = 1.0
    holes: bool = False
