"""FrameTransformer against golden vectors from the reference class (tests/golden/make_golden.py::
gen_frame_transformer: with and without frame offsets).  CPU: the oracle's combine / subtract frame transforms on the
reference's index tables.  GPU: the mirror / mdp_frame_transformer_update, 1e-5 relative, 1e-6 absolute; and the
"nothing outdated -> no work" gate."""
import os

import numpy as np
import pytest
import torch

from conftest import assert_close, to_np
from oracle import math_np as M

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "frame_transformer.npz"))
NF = 4


def _oracle(offsets: bool):
    tr = G["transforms"][G["per_env_indices"]]
    tr = np.concatenate([tr[:, :3], tr[:, [6, 3, 4, 5]]], axis=1)
    src = tr[G["source_frame_body_ids"]]
    sp, sq = src[:, :3], src[:, 3:]
    if offsets:
        sp, sq = M.combine_frame_transforms(sp, sq, G["source_offset_pos"], G["source_offset_quat"])
    tgt = tr[G["target_frame_body_ids"]][G["duplicate_frame_indices"]]
    tp, tq = tgt[:, :3], tgt[:, 3:]
    if offsets:
        tp, tq = M.combine_frame_transforms(tp, tq, G["target_offset_pos"], G["target_offset_quat"])
    n = sp.shape[0]
    rp, rq = M.subtract_frame_transforms(np.repeat(sp, NF, axis=0), np.repeat(sq, NF, axis=0), tp, tq)
    return dict(source_pos_w=sp, source_quat_w=sq, target_pos_w=tp.reshape(n, NF, 3), target_quat_w=tq.reshape(n, NF, 4),
                target_pos_source=rp.reshape(n, NF, 3), target_quat_source=rq.reshape(n, NF, 4))


def test_oracle_frame_transformer_against_reference_vectors():
    got = _oracle(True)
    for name, v in got.items():
        np.testing.assert_allclose(v, G[name], rtol=1e-6, atol=1e-6, err_msg=name)
    got = _oracle(False)
    for name in ("source_pos_w", "target_pos_w", "target_pos_source", "target_quat_source"):
        np.testing.assert_allclose(got[name], G["nooff_" + name], rtol=1e-6, atol=1e-6, err_msg=name)


@pytest.mark.gpu
def test_gpu_frame_transformer_against_reference_vectors(device):
    from isaaclab_amd.sensors import FrameTransformer, FrameTransformerCfg

    n = G["source_pos_w"].shape[0]
    tr = torch.from_numpy(G["transforms"]).to(device)
    view = type("View", (), {"get_transforms": lambda self: tr})()
    tables = {k: G[k] for k in ("per_env_indices", "source_frame_body_ids", "target_frame_body_ids", "duplicate_frame_indices")}
    tables["target_frame_names"] = ["a", "b", "c", "d"]
    for with_offsets in (True, False):
        t = dict(tables)
        if with_offsets:
            t.update(source_frame_offset_pos=G["source_offset_pos"], source_frame_offset_quat=G["source_offset_quat"],
                     target_frame_offset_pos=G["target_offset_pos"], target_frame_offset_quat=G["target_offset_quat"])
        ft = FrameTransformer(FrameTransformerCfg(tables=t), view, n, str(device), 0.005)
        ft.update(0.005)
        d = ft.data
        names = ("source_pos_w", "source_quat_w", "target_pos_w", "target_quat_w", "target_pos_source", "target_quat_source") \
            if with_offsets else ("source_pos_w", "target_pos_w", "target_pos_source", "target_quat_source")
        for name in names:
            assert_close(to_np(getattr(d, name)), G[("" if with_offsets else "nooff_") + name], rtol=1e-5, atol=1e-6, msg=name)
        assert not ft._is_outdated.any() and torch.equal(ft._timestamp_last_update, ft._timestamp)
        # nothing outdated: a later access does not recompute (the body moved, the data must not)
        before = d.target_pos_w.clone()
        tr[:, 0] += 1.0
        assert torch.equal(ft.data.target_pos_w, before)
        ft._is_outdated[3] = True                       # one outdated env: everything is recomputed, like the reference
        assert_close(to_np(ft.data.target_pos_w[:, :, 0]), to_np(before[:, :, 0]) + 1.0, rtol=1e-5, atol=1e-5)
        tr[:, 0] -= 1.0
