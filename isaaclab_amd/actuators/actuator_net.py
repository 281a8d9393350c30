"""Learned actuator models (``LAB/actuators/actuator_net.py:30-188``) on the matrix cores.

The TorchScript file named by ``cfg.network_file`` is only read for its *weights*: the LSTM stack is
taken from ``network.lstm.state_dict()`` exactly as the reference inspects it (actuator_net.py:53-54)
and every other ``Linear`` layer (with the activations between them) forms the decoder / MLP.  The
weights are packed once into MFMA A-fragment order (``mdp_net_pack``) and kept on the device.
"""
from __future__ import annotations

import ctypes as C
import io
from collections.abc import Sequence

import torch

from .. import _lib as L
from ..utils.types import ArticulationActions
from .actuator_pd import DCMotor

_ACT_IDS = {"Identity": 0, "ELU": 1, "ReLU": 2, "Tanh": 3, "Softsign": 4, "Sigmoid": 5}


def _read_network(network_file) -> torch.jit.ScriptModule | torch.nn.Module:
    if isinstance(network_file, (torch.nn.Module, torch.jit.ScriptModule)):
        return network_file
    if isinstance(network_file, (bytes, bytearray)):
        network_file = io.BytesIO(network_file)
    return torch.jit.load(network_file, map_location="cpu")


def _module_kind(m) -> str:
    return getattr(m, "original_name", None) or type(m).__name__


def _extract(network, expect_lstm: bool):
    """-> (lstm_layers, hidden, lstm_input, [lstm tensors], [(W, b, act_id), ...])"""
    lstm_tensors, n_layers, hidden, lstm_in = [], 0, 0, 0
    if expect_lstm:
        sd = network.lstm.state_dict()
        n_layers = len(sd) // 4
        hidden = sd["weight_hh_l0"].shape[1]
        lstm_in = sd["weight_ih_l0"].shape[1]
        for l in range(n_layers):
            for key in (f"weight_ih_l{l}", f"weight_hh_l{l}", f"bias_ih_l{l}", f"bias_hh_l{l}"):
                lstm_tensors.append(sd[key].detach().float().cpu().contiguous())
    linears = []
    for name, mod in network.named_modules():
        kind = _module_kind(mod)
        if kind == "Linear":
            sd = mod.state_dict()
            w = sd["weight"].detach().float().cpu().contiguous()
            b = sd["bias"].detach().float().cpu().contiguous() if "bias" in sd else torch.zeros(w.shape[0])
            linears.append([w, b, 0])
        elif kind in _ACT_IDS and linears:
            linears[-1][2] = _ACT_IDS[kind]
        elif kind in ("LSTM", "Sequential", "RecursiveScriptModule", "ModuleList") or mod is network:
            continue
        elif kind not in _ACT_IDS:
            children = list(mod.children()) if hasattr(mod, "children") else []
            if not children:
                raise NotImplementedError(
                    f"actuator network layer '{name}' of type {kind} has no HIP lowering "
                    f"(supported: LSTM, Linear, {sorted(_ACT_IDS)})"
                )
    if not linears:
        raise NotImplementedError("actuator network without a Linear output layer")
    return n_layers, hidden, lstm_in, lstm_tensors, linears


class _PackedNet:
    def __init__(self, network, expect_lstm: bool, device):
        n_layers, hidden, lstm_in, lstm_tensors, linears = _extract(network, expect_lstm)
        desc = L.MdpNetDesc()
        desc.lstm_layers, desc.lstm_hidden, desc.lstm_input = n_layers, hidden, lstm_in
        desc.mlp_layers = len(linears)
        dims = [linears[0][0].shape[1]] + [w.shape[0] for w, _, _ in linears]
        for i, d in enumerate(dims):
            desc.mlp_dims[i] = d
        for i, (_, _, a) in enumerate(linears):
            desc.mlp_act[i] = a
        size = L.lib.mdp_net_packed_size(C.byref(desc))
        if size < 0:
            raise NotImplementedError(L.last_error())
        tensors = lstm_tensors + [t for w, b, _ in linears for t in (w, b)]
        ptrs = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        packed = torch.empty(size, dtype=torch.float32)
        L.check(L.lib.mdp_net_pack(C.byref(desc), C.byref(ptrs), packed.data_ptr()))
        self.desc = desc
        self.packed = packed.to(device)
        self.num_layers, self.hidden = n_layers, hidden


class ActuatorNetLSTM(DCMotor):
    """actuator_net.py:30-99"""

    def __init__(self, cfg, *args, **kwargs):
        super().__init__(cfg, *args, **kwargs)
        self.network = _read_network(self.cfg.network_file)
        self._net = _PackedNet(self.network, True, self._device)
        num_layers, hidden_dim = self._net.num_layers, self._net.hidden
        rows = self._num_envs * self.num_joints
        self.sea_input = torch.zeros(rows, 1, 2, device=self._device)
        self.sea_hidden_state = torch.zeros(num_layers, rows, hidden_dim, device=self._device)
        self.sea_cell_state = torch.zeros(num_layers, rows, hidden_dim, device=self._device)
        shape = (num_layers, self._num_envs, self.num_joints, hidden_dim)
        self.sea_hidden_state_per_env = self.sea_hidden_state.view(shape)
        self.sea_cell_state_per_env = self.sea_cell_state.view(shape)

    def supports_recast(self) -> bool:
        """The register-resident kernel's shape (hidden 8, <= 2 layers, linear decoder): its gated launch can carry the
        flagged height-scan recast (``mdp_actuator_lstm_compute_gated_with_recast``)."""
        n = self._net
        return n.hidden == 8 and n.num_layers <= 2 and int(n.desc.mlp_layers) == 1

    def reset(self, env_ids: Sequence[int]):
        from ..utils.buffers.circular_buffer import _mask_from_ids

        mask = _mask_from_ids(env_ids, self._num_envs, self._device)
        L.check(L.lib.mdp_actuator_lstm_reset(self._net.num_layers, self._net.hidden, self._num_envs, self.num_joints,
                                              self.sea_hidden_state.data_ptr(), self.sea_cell_state.data_ptr(),
                                              L.ptr(mask) if mask is not None else None, L.stream()))

    def compute(self, control_action, joint_pos, joint_vel):
        q, ld_q = L.rows(joint_pos, "joint_pos")
        qd, ld_qd = L.rows(joint_vel, "joint_vel")
        des, ld_des = L.rows(control_action.joint_positions, "joint_positions")
        gate, self.pending_gate = self.pending_gate, None
        recast, self.pending_recast = self.pending_recast, None
        if gate is not None and recast is not None:
            # the flagged recast of the reset envs rides in front of the actuator blocks (one launch for the reset tail)
            cast, reset_mask = recast
            L.check(L.lib.mdp_actuator_lstm_compute_gated_with_recast(
                gate.data_ptr(), C.byref(self._net.desc), self._net.packed.data_ptr(), self._num_envs, self.num_joints,
                des.data_ptr(), ld_des, q.data_ptr(), ld_q, qd.data_ptr(), ld_qd, self.sea_hidden_state.data_ptr(),
                self.sea_cell_state.data_ptr(), self.effort_limit.data_ptr(), self.velocity_limit.data_ptr(),
                float(self._saturation_effort), self.computed_effort.data_ptr(), self.applied_effort.data_ptr(),
                C.byref(cast), reset_mask.data_ptr(), L.stream(q.device)))
            control_action.joint_efforts = self.applied_effort
            control_action.joint_positions = None
            control_action.joint_velocities = None
            return control_action
        if gate is not None:
            L.check(L.lib.mdp_actuator_lstm_compute_gated(
                gate.data_ptr(), C.byref(self._net.desc), self._net.packed.data_ptr(), self._num_envs, self.num_joints,
                des.data_ptr(), ld_des, q.data_ptr(), ld_q, qd.data_ptr(), ld_qd, self.sea_hidden_state.data_ptr(),
                self.sea_cell_state.data_ptr(), self.effort_limit.data_ptr(), self.velocity_limit.data_ptr(),
                float(self._saturation_effort), self.computed_effort.data_ptr(), self.applied_effort.data_ptr(),
                L.stream(q.device)))
            control_action.joint_efforts = self.applied_effort
            control_action.joint_positions = None
            control_action.joint_velocities = None
            return control_action
        act = self._take_joint_action(des.data_ptr(), ld_des)
        if act is not None:
            L.check(L.lib.mdp_actuator_lstm_compute_with_action(
                C.byref(act), C.byref(self._net.desc), self._net.packed.data_ptr(), self._num_envs, self.num_joints,
                q.data_ptr(), ld_q, qd.data_ptr(), ld_qd, self.sea_hidden_state.data_ptr(),
                self.sea_cell_state.data_ptr(), self.effort_limit.data_ptr(), self.velocity_limit.data_ptr(),
                float(self._saturation_effort), self.computed_effort.data_ptr(), self.applied_effort.data_ptr(),
                L.stream(q.device)))
            control_action.joint_efforts = self.applied_effort
            control_action.joint_positions = None
            control_action.joint_velocities = None
            return control_action
        upd, self.pending_scene_update = self.pending_scene_update, None
        L.check(L.lib.mdp_actuator_lstm_compute_after_update(
            C.byref(upd[0]) if upd is not None else None, float(upd[1]) if upd is not None else 0.0,
            C.byref(self._net.desc), self._net.packed.data_ptr(), self._num_envs, self.num_joints,
            des.data_ptr(), ld_des, q.data_ptr(), ld_q, qd.data_ptr(), ld_qd,
            self.sea_hidden_state.data_ptr(), self.sea_cell_state.data_ptr(),
            self.effort_limit.data_ptr(), self.velocity_limit.data_ptr(), float(self._saturation_effort),
            self.computed_effort.data_ptr(), self.applied_effort.data_ptr(), L.stream(q.device)))
        control_action.joint_efforts = self.applied_effort
        control_action.joint_positions = None
        control_action.joint_velocities = None
        return control_action


class ActuatorNetMLP(DCMotor):
    """actuator_net.py:102-188"""

    carries_scene_update = False   # own kernel (history shift): the update is launched separately
    folds_joint_action = False
    supports_gate = False

    def __init__(self, cfg, *args, **kwargs):
        super().__init__(cfg, *args, **kwargs)
        if self.cfg.input_order not in ("pos_vel", "vel_pos"):
            raise ValueError(
                f"Invalid input order for MLP actuator net: {self.cfg.input_order}. Must be 'pos_vel' or 'vel_pos'."
            )
        self.network = _read_network(self.cfg.network_file)
        self._net = _PackedNet(self.network, False, self._device)
        self._input_idx = torch.tensor(list(self.cfg.input_idx), dtype=torch.int32, device=self._device)
        history_length = max(self.cfg.input_idx) + 1
        self._history_length = history_length
        self._joint_pos_error_history = torch.zeros(self._num_envs, history_length, self.num_joints, device=self._device)
        self._joint_vel_history = torch.zeros(self._num_envs, history_length, self.num_joints, device=self._device)

    def reset(self, env_ids: Sequence[int]):
        from ..utils.buffers.circular_buffer import _mask_from_ids

        mask = _mask_from_ids(env_ids, self._num_envs, self._device)
        per_env = self._history_length * self.num_joints
        for buf in (self._joint_pos_error_history, self._joint_vel_history):
            L.check(L.lib.mdp_rows_zero(buf.data_ptr(), self._num_envs, per_env,
                                        L.ptr(mask) if mask is not None else None, L.stream()))

    def compute(self, control_action, joint_pos, joint_vel):
        q, ld_q = L.rows(joint_pos, "joint_pos")
        qd, ld_qd = L.rows(joint_vel, "joint_vel")
        des, ld_des = L.rows(control_action.joint_positions, "joint_positions")
        L.check(L.lib.mdp_actuator_mlp_compute(
            C.byref(self._net.desc), self._net.packed.data_ptr(), self._num_envs, self.num_joints,
            self._history_length, des.data_ptr(), ld_des, q.data_ptr(), ld_q, qd.data_ptr(), ld_qd,
            self._joint_pos_error_history.data_ptr(), self._joint_vel_history.data_ptr(),
            self._input_idx.data_ptr(), int(self._input_idx.numel()), 0 if self.cfg.input_order == "pos_vel" else 1,
            float(self.cfg.pos_scale), float(self.cfg.vel_scale), float(self.cfg.torque_scale),
            self.effort_limit.data_ptr(), self.velocity_limit.data_ptr(), float(self._saturation_effort),
            self.computed_effort.data_ptr(), self.applied_effort.data_ptr(), L.stream(q.device)))
        control_action.joint_efforts = self.applied_effort
        control_action.joint_positions = None
        control_action.joint_velocities = None
        return control_action
