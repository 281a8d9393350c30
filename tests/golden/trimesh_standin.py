"""Functional stand-in for the handful of ``trimesh`` calls the reference terrain generator makes
(``trimesh.creation.box``, ``trimesh.transformations.translation_matrix``, ``trimesh.util.concatenate``,
``Trimesh(vertices, faces)`` with ``apply_transform / apply_translation / update_faces / triangles``).  trimesh itself
is not installed and not part of /root/reference, so the golden vectors of tests/golden/mesh_terrains.npz pin everything
the REFERENCE computes (box sizes, positions, order, step counts, random draws, layout, border, centring) on top of this
restatement of trimesh's published box layout -- the box primitive's own vertex / face order is "parity unpinned".
Test infrastructure only: used by make_golden.py, never imported by the product."""
import types

import numpy as np


class Trimesh:
    def __init__(self, vertices=None, faces=None, process=True, **kwargs):
        self.vertices = np.array(vertices, dtype=np.float64).reshape(-1, 3)
        self.faces = np.array(faces, dtype=np.int64).reshape(-1, 3)

    @property
    def triangles(self):
        return self.vertices[self.faces]

    def apply_transform(self, matrix):
        m = np.asarray(matrix, dtype=np.float64)
        stack = np.column_stack((self.vertices, np.ones(len(self.vertices))))
        self.vertices = np.dot(m, stack.T).T[:, :3]
        return self

    def apply_translation(self, t):
        m = np.eye(4)
        m[:3, 3] = t
        return self.apply_transform(m)

    def update_faces(self, mask):
        self.faces = self.faces[np.asarray(mask)]


def _box(extents=None, transform=None, **kwargs):
    v = np.array([0, 0, 0, 0, 0, 1, 0, 1, 0, 0, 1, 1, 1, 0, 0, 1, 0, 1, 1, 1, 0, 1, 1, 1], dtype=np.float64).reshape(-1, 3)
    v -= 0.5
    if extents is not None:
        v *= np.asarray(extents, dtype=np.float64)
    f = np.array([1, 3, 0, 4, 1, 0, 0, 3, 2, 2, 4, 0, 1, 7, 3, 5, 1, 4, 5, 7, 1, 3, 7, 2, 6, 4, 2, 2, 7, 6, 6, 5, 4, 7, 5, 6],
                 dtype=np.int64).reshape(-1, 3)
    m = Trimesh(v, f, process=False)
    if transform is not None:
        m.apply_transform(transform)
    return m


def _translation_matrix(direction):
    m = np.identity(4)
    m[:3, 3] = direction[:3]
    return m


def _concatenate(a, b=None):
    meshes = list(a) if isinstance(a, (list, tuple)) else [a]
    if b is not None:
        meshes += list(b) if isinstance(b, (list, tuple)) else [b]
    off, vs, fs = 0, [], []
    for m in meshes:
        vs.append(np.asarray(m.vertices, np.float64).reshape(-1, 3))
        fs.append(np.asarray(m.faces, np.int64).reshape(-1, 3) + off)
        off += len(vs[-1])
    return Trimesh(np.concatenate(vs), np.concatenate(fs), process=False)


def module() -> types.ModuleType:
    mod = types.ModuleType("trimesh")
    mod.Trimesh = Trimesh
    mod.creation = types.SimpleNamespace(box=_box)
    mod.transformations = types.SimpleNamespace(translation_matrix=_translation_matrix)
    mod.util = types.SimpleNamespace(concatenate=_concatenate)
    return mod
