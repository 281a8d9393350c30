"""ctypes wrapper of ``oracle/raycast_ref.c`` (test infrastructure; a self-contained restatement of the published
watertight ray/triangle test -- it shares no source with ``isaaclab_amd/csrc``)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libraycast_ref.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        _LIB = C.CDLL(path)
        _LIB.rcref_bins_build.restype = C.c_void_p
        _LIB.rcref_bins_build.argtypes = [C.c_void_p, C.c_int32, C.c_float]
        _LIB.rcref_bins_free.argtypes = [C.c_void_p]
        _LIB.rcref_brute.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_float,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB.rcref_bins_cast.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return _LIB


def tri_soup(points: np.ndarray, indices: np.ndarray) -> np.ndarray:
    pts = np.asarray(points, dtype=np.float32).reshape(-1, 3)
    idx = np.asarray(indices, dtype=np.int64).reshape(-1, 3)
    return np.ascontiguousarray(pts[idx].reshape(-1, 9))


def _outputs(n):
    return (np.empty((n, 3), np.float32), np.empty(n, np.float32), np.empty((n, 3), np.float32),
            np.empty(n, np.int32))


def raycast_brute(soup: np.ndarray, starts: np.ndarray, dirs: np.ndarray, max_dist: float = 1e6):
    s = np.ascontiguousarray(starts, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
    hits, dist, normal, face = _outputs(s.shape[0])
    _lib().rcref_brute(soup.ctypes.data, soup.shape[0], s.shape[0], s.ctypes.data, d.ctypes.data, max_dist,
                                hits.ctypes.data, dist.ctypes.data, normal.ctypes.data, face.ctypes.data)
    return hits, dist, normal, face


class GridCaster:
    def __init__(self, soup: np.ndarray, cell: float | None = None):
        self.soup = np.ascontiguousarray(soup, dtype=np.float32)
        if cell is None:
            ex = np.ptp(self.soup.reshape(-1, 3)[:, 0]) + 1e-6
            ey = np.ptp(self.soup.reshape(-1, 3)[:, 1]) + 1e-6
            cell = float(np.sqrt(2.0 * ex * ey / self.soup.shape[0]))
            while (ex / cell + 2) * (ey / cell + 2) > 8e6:
                cell *= 1.5
        self._g = _lib().rcref_bins_build(self.soup.ctypes.data, self.soup.shape[0], cell)

    def cast(self, starts, dirs, max_dist: float = 1e6, num_threads: int = 0):
        s = np.ascontiguousarray(starts, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        hits, dist, normal, face = _outputs(s.shape[0])
        _lib().rcref_bins_cast(self._g, s.shape[0], s.ctypes.data, d.ctypes.data, max_dist, hits.ctypes.data,
                                   dist.ctypes.data, normal.ctypes.data, face.ctypes.data, num_threads)
        return hits, dist, normal, face

    def __del__(self):
        try:
            _lib().rcref_bins_free(self._g)
        except Exception:
            pass
