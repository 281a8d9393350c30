"""CPU: the C-ABI library loads and exports every function include/mdp_hip.h declares; struct mirrors agree
with the header; error reporting works without a GPU (argument validation happens before any HIP call)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from conftest import ROOT

import isaaclab_amd._lib as L

HDR = os.path.join(ROOT, "include", "mdp_hip.h")


def test_every_declared_symbol_is_exported():
    names = L.exported_symbols()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(L.lib, n)]
    assert not missing, f"declared in include/mdp_hip.h but not exported by libmdp_hip.so: {missing}"


def test_version_and_error_string():
    assert L.lib.mdp_version() >= 1
    assert isinstance(L.last_error(), str)


def _c_sizeof(structs):
    """Compile a tiny C program against the public header and print sizeof() of the structs."""
    src = '#include <stdio.h>\n#include "mdp_hip.h"\nint main(){' + "".join(
        f'printf("%zu\\n", sizeof({s}));' for s in structs) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        with open(c, "w") as f:
            f.write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    return [int(v) for v in out]


def test_struct_mirrors_match_header():
    """The header is plain C (compiles with gcc -std=c99) and the ctypes mirrors have the same size."""
    pairs = [("mdp_term_t", L.MdpTerm), ("mdp_raycaster_desc_t", L.MdpRaycasterDesc),
             ("mdp_velocity_command_cfg_t", L.MdpVelocityCommandCfg), ("mdp_net_desc_t", L.MdpNetDesc),
             ("mdp_inhand_cfg_t", L.MdpInhandCfg), ("mdp_scene_update_t", L.MdpSceneUpdate), ("mdp_height_scan_obs_t", L.MdpHeightScanObs),
             ("mdp_termination_reward_args_t", L.MdpTerminationRewardArgs),
             ("mdp_velocity_command_args_t", L.MdpVelocityCommandArgs), ("mdp_obs_args_t", L.MdpObsArgs),
             ("mdp_raycaster_update_args_t", L.MdpRaycasterUpdateArgs),
             ("mdp_joint_action_t", L.MdpJointAction), ("mdp_fill_t", L.MdpFill), ("mdp_post_reset_args_t", L.MdpPostResetArgs), ("mdp_post_events_t", L.MdpPostEvents),
             ("mdp_event_root_state_t", L.MdpEventRootState), ("mdp_event_joints_t", L.MdpEventJoints),
             ("mdp_event_push_t", L.MdpEventPush), ("mdp_curriculum_terrain_t", L.MdpCurriculumTerrain), ("mdp_obs_normalize_t", L.MdpObsNormalize), ("mdp_imu_args_t", L.MdpImuArgs), ("mdp_frame_transformer_args_t", L.MdpFrameTransformerArgs),
             ("mdp_raycaster_camera_args_t", L.MdpRaycasterCameraArgs), ("mdp_event_reset_mark_t", L.MdpEventResetMark),
             ("mdp_inhand_goal_reset_t", L.MdpInhandGoalReset), ("mdp_inhand_obs_args_t", L.MdpInhandObsArgs),
             ("mdp_inhand_reset_t", L.MdpInhandReset), ("mdp_stats_source_t", L.MdpStatsSource)]
    sizes = _c_sizeof([p[0] for p in pairs])
    for (name, cls), sz in zip(pairs, sizes):
        assert C.sizeof(cls) == sz, f"{name}: header {sz} bytes, ctypes mirror {C.sizeof(cls)}"


def test_opcode_tables_are_dense():
    for prefix, count in (("MDP_OBS_", "MDP_OBS_NUM_OPS"), ("MDP_REW_", "MDP_REW_NUM_OPS"), ("MDP_TERM_", "MDP_TERM_NUM_OPS")):
        ids = sorted(v for k, v in L.OPS.items() if k.startswith(prefix) and not k.endswith("NUM_OPS"))
        assert ids == list(range(L.OPS[count])), prefix


def test_header_cites_reference_lines():
    text = open(HDR).read()
    protos = re.findall(r"\n(?:int|int64_t|const char\s*\*)\s+(mdp_\w+)\s*\(", text)
    assert len(protos) >= 40
    # every prototype is preceded by a comment naming the reference interface it replaces (file:line)
    assert len(re.findall(r"[\w/]+\.py:\d+", text)) >= len(protos)


def test_invalid_arguments_fail_without_touching_the_gpu():
    # N < 0 -> MDP_ERR_INVALID before any HIP call
    rc = L.lib.mdp_actuator_pd_compute(0, -1, 12, None, 0, None, 0, None, 0, None, 0, None, 0, None, None, None, None,
                                       0.0, None, None, None)
    assert rc == L.MDP_ERR_INVALID
    assert L.last_error()
    with pytest.raises(ValueError):
        L.check(rc)


def test_product_path_has_no_cpu_fallback():
    import torch

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.require_gpu(torch.zeros(3))
    # the package never imports the oracle
    import isaaclab_amd  # noqa: F401
    pkg = os.path.join(ROOT, "isaaclab_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, fn)


def test_net_pack_fragment_tables_and_register_kernel_section():
    """mdp_net_pack is host code (no GPU): for the ANYdrive shape (LSTM(2, 8, 2) + Linear(8, 1)) the blob holds the
    MFMA fragment tables and, behind them, the register kernel's section -- the same numbers, four fragments of a lane
    side by side in the order the kernel unpacks them (per layer: weight fragments k-step major, eight bias fragments;
    then the decoder's two weight and four bias fragments).  Other shapes have no such section."""
    import numpy as np

    def pack(hidden, layers):
        rng = np.random.default_rng(3)
        desc = L.MdpNetDesc()
        desc.lstm_layers, desc.lstm_hidden, desc.lstm_input, desc.mlp_layers = layers, hidden, 2, 1
        desc.mlp_dims[0], desc.mlp_dims[1] = hidden, 1
        arrs = []
        for l in range(layers):
            d_in = 2 if l == 0 else hidden
            arrs += [rng.standard_normal((4 * hidden, d_in)), rng.standard_normal((4 * hidden, hidden)),
                     rng.standard_normal(4 * hidden), rng.standard_normal(4 * hidden)]
        arrs += [rng.standard_normal((1, hidden)), rng.standard_normal(1)]
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in arrs]
        size = L.lib.mdp_net_packed_size(C.byref(desc))
        assert size > 0
        out = np.full(size, np.nan, dtype=np.float32)
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        L.check(L.lib.mdp_net_pack(C.byref(desc), C.byref(ptrs), out.ctypes.data))
        assert np.isfinite(out).all()
        return out

    tpl = 2
    for layers, n_frag in ((1, 20), (2, 36)):
        out = pack(8, layers)
        tables = sum(((1 if l == 0 else tpl) + tpl) * tpl * 64 + 4 * tpl * 64 for l in range(layers)) + tpl * 2 * 64 + 2 * 4 * 64
        assert out.size == tables + n_frag * 64
        src, off = [], 0
        for l in range(layers):
            nk = (1 if l == 0 else tpl) + tpl
            src += [off + (k * tpl + t) * 64 for k in range(nk) for t in range(tpl)]
            off += nk * tpl * 64
            src += [off + i * 64 for i in range(4 * tpl)]
            off += 4 * tpl * 64
        src += [off + (s * 2) * 64 for s in range(tpl)]
        off += tpl * 2 * 64
        src += [off + r * 64 for r in range(4)]
        assert len(src) == n_frag
        sec = out[tables:].reshape(n_frag // 4, 64, 4)
        for i, o in enumerate(src):
            np.testing.assert_array_equal(sec[i // 4, :, i % 4], out[o:o + 64])
    # hidden 16: fragment tables only (LDS kernel)
    out16 = pack(16, 1)
    tpl16 = 4
    assert out16.size == (1 + tpl16) * tpl16 * 64 + 4 * tpl16 * 64 + tpl16 * 2 * 64 + 2 * 4 * 64
