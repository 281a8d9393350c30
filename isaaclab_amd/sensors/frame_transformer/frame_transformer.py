"""``FrameTransformer`` (``LAB/sensors/frame_transformer/frame_transformer.py:30-385``), the per-step part: poses of the
target frames in the world and relative to the source frame (pure math over the body transforms).

The reference resolves its index tables by walking the USD stage at initialisation (:133-320); there is no stage here,
so the constructor takes the resolved tables (``FrameTransformerCfg.tables``) -- same names as the reference's private
attributes.  The per-step fill is one launch (``mdp_frame_transformer_update``)."""
from __future__ import annotations

import ctypes as C
from collections.abc import Sequence
from dataclasses import dataclass

import torch

from ... import _lib as L
from ...utils.configclass import configclass
from ..sensor_base import SensorBase, SensorBaseCfg


@dataclass
class FrameTransformerData:
    """frame_transformer_data.py:13-60"""

    target_frame_names: list | None = None
    target_pos_source: torch.Tensor | None = None
    target_quat_source: torch.Tensor | None = None
    target_pos_w: torch.Tensor | None = None
    target_quat_w: torch.Tensor | None = None
    source_pos_w: torch.Tensor | None = None
    source_quat_w: torch.Tensor | None = None


class FrameTransformer(SensorBase):
    def __init__(self, cfg: "FrameTransformerCfg", frame_view, num_envs: int, device: str, sim_physics_dt: float):
        super().__init__(cfg, num_envs, device, sim_physics_dt)
        self._frame_physx_view = frame_view
        t = cfg.tables
        i64 = lambda x: torch.as_tensor(x, dtype=torch.int64, device=device).contiguous()
        f32 = lambda x: None if x is None else torch.as_tensor(x, dtype=torch.float32, device=device).contiguous()
        self._target_frame_names = list(t["target_frame_names"])
        self._per_env_indices = i64(t["per_env_indices"])
        self._source_frame_body_ids = i64(t["source_frame_body_ids"])
        self._target_frame_body_ids = i64(t["target_frame_body_ids"])
        self._duplicate_frame_indices = i64(t["duplicate_frame_indices"])
        self._source_frame_offset_pos, self._source_frame_offset_quat = f32(t.get("source_frame_offset_pos")), f32(t.get("source_frame_offset_quat"))
        self._target_frame_offset_pos, self._target_frame_offset_quat = f32(t.get("target_frame_offset_pos")), f32(t.get("target_frame_offset_quat"))
        nf = len(self._target_frame_names)
        if self._source_frame_body_ids.shape != (num_envs,) or self._duplicate_frame_indices.shape != (num_envs * nf,):
            raise ValueError("index tables do not match num_envs / the number of target frames")
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=device)
        self._data = FrameTransformerData(
            target_frame_names=self._target_frame_names, source_pos_w=z(num_envs, 3), source_quat_w=z(num_envs, 4),
            target_pos_w=z(num_envs, nf, 3), target_quat_w=z(num_envs, nf, 4), target_pos_source=z(num_envs, nf, 3),
            target_quat_source=z(num_envs, nf, 4))

    @property
    def data(self) -> FrameTransformerData:
        self._update_outdated_buffers()
        return self._data

    @property
    def num_bodies(self) -> int:
        return len(self._target_frame_names)

    @property
    def body_names(self) -> list:
        return self._target_frame_names

    def reset(self, env_ids: Sequence[int] | None = None):
        super().reset(env_ids)

    def update(self, dt: float, force_recompute: bool = False):
        L.check(L.lib.mdp_sensor_tick(self._num_envs, L.ptr(self._timestamp), L.ptr(self._timestamp_last_update),
                                      L.ptr(self._is_outdated), float(dt), float(self.cfg.update_period), L.stream()))
        if force_recompute or self._is_visualizing or (self.cfg.history_length > 0):
            self._update_outdated_buffers()

    def _update_outdated_buffers(self):
        tr = self._frame_physx_view.get_transforms()
        if tr.dim() != 2 or tr.shape[1] != 7 or not tr.is_contiguous() or tr.dtype != torch.float32:
            raise ValueError("frame view transforms must be a contiguous float32 (M, 7) tensor")
        gate = self._is_outdated.sum(dtype=torch.float32).reshape(1)       # on the device: nothing outdated -> no work
        a = L.MdpFrameTransformerArgs()
        a.gate, a.is_outdated = gate.data_ptr(), self._is_outdated.data_ptr()
        a.timestamp, a.timestamp_last_update = self._timestamp.data_ptr(), self._timestamp_last_update.data_ptr()
        a.transforms, a.per_env_indices = tr.data_ptr(), self._per_env_indices.data_ptr()
        a.source_frame_body_ids, a.target_frame_body_ids = self._source_frame_body_ids.data_ptr(), self._target_frame_body_ids.data_ptr()
        a.duplicate_frame_indices = self._duplicate_frame_indices.data_ptr()
        for name in ("source", "target"):
            pos, quat = getattr(self, f"_{name}_frame_offset_pos"), getattr(self, f"_{name}_frame_offset_quat")
            setattr(a, f"{name}_offset_pos", None if pos is None else pos.data_ptr())
            setattr(a, f"{name}_offset_quat", None if quat is None else quat.data_ptr())
        d = self._data
        a.source_pos_w, a.source_quat_w = d.source_pos_w.data_ptr(), d.source_quat_w.data_ptr()
        a.target_pos_w, a.target_quat_w = d.target_pos_w.data_ptr(), d.target_quat_w.data_ptr()
        a.target_pos_source, a.target_quat_source = d.target_pos_source.data_ptr(), d.target_quat_source.data_ptr()
        L.check(L.lib.mdp_frame_transformer_update(self._num_envs, len(self._target_frame_names), C.byref(a), L.stream(self._device)))
        self._keep = gate


@configclass
class FrameTransformerCfg(SensorBaseCfg):
    class_type: type = FrameTransformer
    #: the resolved tables: target_frame_names, per_env_indices, source_frame_body_ids, target_frame_body_ids,
    #: duplicate_frame_indices and the optional {source,target}_frame_offset_{pos,quat}
    tables: dict = dict()
