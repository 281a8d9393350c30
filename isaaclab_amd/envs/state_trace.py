"""State-trace wire format + replay (first slice).

A trace is what the MDP hot path reads from physics, recorded once per physics sub-step: the buffers the PhysX tensor
views hand out (``get_root_transforms`` / ``get_root_velocities`` / ``get_dof_positions`` / ``get_dof_velocities`` /
``get_link_transforms`` / ``get_link_velocities`` / ``get_link_accelerations`` / ``get_net_contact_forces``;
articulation_data.py:273-441, contact_sensor.py:323), laid out exactly like one frame of ``SyntheticPhysicsView``,
plus the actions of every env step.  Recorded from a real Isaac Sim run (``TraceWriter.append`` after each
``sim.step()``) or from the synthetic generator, it replays through ``ManagerBasedRLEnv`` unchanged, so parity can be
checked on real trajectories.

File layout (little endian): ``b"MDPTRACE"`` | u32 version | u32 header length | JSON header (utf-8) | padding to 64 B |
frames ``(num_frames, frame_size)`` f32 | actions ``(num_steps, num_envs, action_dim)`` f32.
Frame 0 is the state before the first step; frame ``1 + k * decimation + i`` is sub-step ``i`` of env step ``k``.
The frames section can be memory-mapped (``load_trace(path, mmap=True)``): long traces are streamed, not loaded."""
from __future__ import annotations

import json
import struct

import numpy as np
import torch

MAGIC = b"MDPTRACE"
VERSION = 1
_ALIGN = 64


def _header(rec, num_frames: int, num_steps: int, action_dim: int) -> dict:
    return {
        "name": rec.name, "num_envs": int(rec.num_envs), "decimation": int(rec.decimation),
        "frame_size": int(rec.frames.shape[1]) if rec.frames is not None else int(rec._frame_size),
        "num_frames": int(num_frames), "num_steps": int(num_steps), "action_dim": int(action_dim),
        "fields": {k: [int(o), int(s), [int(d) for d in shape]] for k, (o, s, shape) in rec.fields.items()},
        "env_origins": None if rec.env_origins is None else np.asarray(rec.env_origins, np.float32).tolist(),
        "quaternion_order": "xyzw", "dtype": "f32",
    }


def _write_header(f, header: dict) -> int:
    blob = json.dumps(header, separators=(",", ":")).encode("utf-8")
    f.write(MAGIC + struct.pack("<II", VERSION, len(blob)) + blob)
    pad = (-f.tell()) % _ALIGN
    f.write(b"\0" * pad)
    return f.tell()


def save_trace(path: str, rec) -> None:
    """Writes a ``Recording`` (frames + actions + field table).  Random-number injections and terrain meshes are not part
    of a trace: they belong to the run that replays it."""
    frames = rec.frames.detach().cpu().to(torch.float32).contiguous().numpy()
    actions = rec.actions.detach().cpu().to(torch.float32).contiguous().numpy()
    if frames.ndim != 2 or actions.ndim != 3 or actions.shape[1] != rec.num_envs:
        raise ValueError(f"bad trace shapes: frames {frames.shape}, actions {actions.shape}")
    if frames.shape[0] != 1 + actions.shape[0] * rec.decimation:
        raise ValueError(f"{frames.shape[0]} frames do not match {actions.shape[0]} steps x decimation {rec.decimation} (+ 1)")
    with open(path, "wb") as f:
        _write_header(f, _header(rec, frames.shape[0], actions.shape[0], actions.shape[2]))
        f.write(frames.tobytes())
        f.write(actions.tobytes())


def read_header(path: str) -> tuple[dict, int]:
    """-> (header, byte offset of the frames section).  Raises ``ValueError`` for anything that is not a trace."""
    with open(path, "rb") as f:
        head = f.read(16)
        if len(head) < 16 or head[:8] != MAGIC:
            raise ValueError(f"{path}: not a state trace (bad magic)")
        version, hlen = struct.unpack("<II", head[8:16])
        if version != VERSION:
            raise ValueError(f"{path}: trace version {version}, this reader knows {VERSION}")
        blob = f.read(hlen)
        if len(blob) != hlen:
            raise ValueError(f"{path}: truncated header")
        header = json.loads(blob.decode("utf-8"))
        pos = f.tell()
    return header, pos + ((-pos) % _ALIGN)


def load_trace(path: str, mmap: bool = False):
    """-> ``Recording`` with ``frames`` / ``actions`` / ``fields`` / ``env_origins`` filled.  ``mmap``: the frames stay on
    disk behind a read-only memory map (slice them into ``SyntheticPhysicsView.set_frames`` window by window)."""
    import os

    from .synthetic import Recording

    h, off = read_header(path)
    n_f, fs = h["num_frames"], h["frame_size"]
    n_a = h["num_steps"] * h["num_envs"] * h["action_dim"]
    need = off + 4 * (n_f * fs + n_a)
    if os.path.getsize(path) < need:
        raise ValueError(f"{path}: truncated trace ({os.path.getsize(path)} bytes, header promises {need})")
    if mmap:
        frames = np.memmap(path, dtype="<f4", mode="r", offset=off, shape=(n_f, fs))
    else:
        frames = np.fromfile(path, dtype="<f4", count=n_f * fs, offset=off).reshape(n_f, fs)
    actions = np.fromfile(path, dtype="<f4", count=n_a, offset=off + 4 * n_f * fs).reshape(h["num_steps"], h["num_envs"], h["action_dim"])
    rec = Recording()
    rec.name, rec.num_envs, rec.decimation = h["name"], h["num_envs"], h["decimation"]
    rec.fields = {k: (o, s, tuple(shape)) for k, (o, s, shape) in h["fields"].items()}
    rec.frames = frames if mmap else torch.from_numpy(frames)
    rec.actions = torch.from_numpy(actions)
    rec.env_origins = None if h["env_origins"] is None else np.asarray(h["env_origins"], np.float32)
    return rec


class TraceWriter:
    """Incremental writer: ``append_frame(view)`` after every physics step copies the view's current buffers (any object
    with the PhysX getter names), ``append_actions(a)`` once per env step; ``close()`` finalises the header."""

    GETTERS = {"root_transforms": "get_root_transforms", "root_velocities": "get_root_velocities",
               "dof_positions": "get_dof_positions", "dof_velocities": "get_dof_velocities",
               "link_transforms": "get_link_transforms", "link_velocities": "get_link_velocities",
               "link_accelerations": "get_link_accelerations", "net_contact_forces": "get_net_contact_forces"}

    def __init__(self, path: str, name: str, num_envs: int, decimation: int, fields: dict, frame_size: int,
                 env_origins=None):
        from .synthetic import Recording

        self._rec = Recording()
        self._rec.name, self._rec.num_envs, self._rec.decimation = name, num_envs, decimation
        self._rec.fields, self._rec.env_origins, self._rec._frame_size = dict(fields), env_origins, frame_size
        self._path, self._frame_size = path, frame_size
        self._frames_path = path + ".frames.tmp"
        self._frames = open(self._frames_path, "wb")
        self._actions: list[np.ndarray] = []
        self._num_frames = 0

    def append_frame(self, view) -> None:
        frame = np.zeros(self._frame_size, np.float32)
        for name, (o, s, _) in self._rec.fields.items():
            t = getattr(view, self.GETTERS[name])()
            frame[o : o + s] = t.detach().reshape(-1).to(torch.float32).cpu().numpy()
        self._frames.write(frame.tobytes())
        self._num_frames += 1

    def append_actions(self, actions: torch.Tensor) -> None:
        self._actions.append(actions.detach().to(torch.float32).cpu().numpy().copy())

    def close(self) -> None:
        import os
        import shutil

        self._frames.close()
        steps = len(self._actions)
        if self._num_frames != 1 + steps * self._rec.decimation:
            raise ValueError(f"{self._num_frames} frames do not match {steps} steps x decimation {self._rec.decimation} (+ 1)")
        actions = np.stack(self._actions) if steps else np.zeros((0, self._rec.num_envs, 0), np.float32)
        with open(self._path, "wb") as f:
            _write_header(f, _header(self._rec, self._num_frames, steps, actions.shape[2]))
            with open(self._frames_path, "rb") as src:
                shutil.copyfileobj(src, f)
            f.write(actions.tobytes())
        os.remove(self._frames_path)
