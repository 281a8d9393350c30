"""RayCasterCamera class mirror against vectors the reference produced
(tests/golden/make_golden.py::gen_camera_class): pinhole pattern, intrinsics, PinholeCameraPatternCfg.from_intrinsic_matrix,
camera-frame conventions, look-at rotations (CPU), and the class flow on the device -- offsets computed by
set_world_poses / set_world_poses_from_view, set_intrinsic_matrices, one update through mdp_raycaster_camera_update.
fp32: rtol 1e-5, atol 1e-5 (the ray cast behind the reference vectors is the CPU oracle: parity of the cast itself is
unpinned by the reference)."""
import os

import numpy as np
import pytest
import torch

from isaaclab_amd.sensors.ray_caster import patterns
from isaaclab_amd.utils import camera_math as cm

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "camera_class.npz"))
K = [38.0, 0.0, 4.3, 0.0, 35.0, 2.6, 0.0, 0.0, 1.0]
TOL = dict(rtol=1e-5, atol=1e-5)


def pattern_cfg():
    return patterns.PinholeCameraPatternCfg.from_intrinsic_matrix(K, width=8, height=6, focal_length=12.0)


def test_pattern_cfg_from_intrinsic_matrix():
    pc = pattern_cfg()
    got = [pc.focal_length, pc.horizontal_aperture, pc.vertical_aperture, pc.horizontal_aperture_offset,
           pc.vertical_aperture_offset, pc.width, pc.height]
    np.testing.assert_allclose(got, G["pc_fields"], rtol=1e-12)


def test_pinhole_pattern_equals_the_reference():
    starts, dirs = patterns.pinhole_camera_pattern(pattern_cfg(), torch.from_numpy(G["cls_intrinsics"]), "cpu")
    assert starts.shape == dirs.shape == (5, 48, 3) and not starts.any()
    np.testing.assert_allclose(dirs.numpy(), G["cls_ray_directions"], **TOL)
    np.testing.assert_allclose(np.linalg.norm(dirs.numpy(), axis=-1), 1.0, atol=1e-6)


def test_camera_frame_conventions_equal_the_reference():
    q = G["conv_q"]
    for origin in ("opengl", "ros", "world"):
        for target in ("opengl", "ros", "world"):
            got = cm.convert_camera_frame_orientation_convention(q, origin=origin, target=target)
            exp = G[f"conv_{origin}_{target}"]
            # a quaternion and its negative are the same rotation; the branch of quat_from_matrix is the reference's
            np.testing.assert_allclose(got, exp, err_msg=f"{origin}->{target}", **TOL)
    # round trip
    back = cm.convert_camera_frame_orientation_convention(
        cm.convert_camera_frame_orientation_convention(q, "ros", "world"), "world", "ros")
    np.testing.assert_allclose(np.abs((back * q).sum(-1)), 1.0, atol=1e-5)


def test_look_at_rotation_equals_the_reference():
    for up in ("Z", "Y"):
        got = cm.create_rotation_matrix_from_view(G["view_eyes"], G["view_targets"], up_axis=up)
        np.testing.assert_allclose(got, G[f"view_rot_{up}"], err_msg=up, **TOL)
    with pytest.raises(ValueError):
        cm.create_rotation_matrix_from_view(G["view_eyes"], G["view_targets"], up_axis="X")


def test_unsupported_data_types_are_rejected():
    from isaaclab_amd.sensors.ray_caster import RayCasterCamera, RayCasterCameraCfg

    cfg = RayCasterCameraCfg(mesh_prim_paths=["/m"], pattern_cfg=pattern_cfg(), data_types=["rgb"])
    assert cfg.attach_yaw_only is False
    with pytest.raises(ValueError, match="does not support"):
        RayCasterCamera(cfg, None, None, 1, "cpu", 0.005)


@pytest.mark.gpu
@pytest.mark.parametrize("accel", ("grid", "bvh"))
def test_gpu_camera_class_flow(device, accel):
    from isaaclab_amd.sensors.ray_caster import RayCasterCamera, RayCasterCameraCfg
    from isaaclab_amd.utils.warp import convert_to_warp_mesh

    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    mesh = convert_to_warp_mesh(G["vertices"], G["faces"], str(device), accel=accel)
    n = 5
    pose_wxyz = G["cls_view_pose"]
    pose_xyzw = np.concatenate([pose_wxyz[:, :3], pose_wxyz[:, 4:7], pose_wxyz[:, 3:4]], 1)

    class View:                                          # physx.RigidBodyView: (N, 7) [pos, quat xyzw]
        def __init__(self):
            self.t = dev(pose_xyzw)

        def get_transforms(self):
            return self.t

    cfg = RayCasterCameraCfg(mesh_prim_paths=["/m"], pattern_cfg=pattern_cfg(), max_distance=4.0, depth_clipping_behavior="max",
                             data_types=["distance_to_image_plane", "distance_to_camera", "normals"],
                             offset=RayCasterCameraCfg.OffsetCfg(pos=(0.1, -0.05, 0.2), rot=(0.9238795, 0.0, 0.3826834, 0.0), convention="ros"))
    cam = RayCasterCamera(cfg, View(), mesh, n, str(device), 0.005)
    t = lambda x: x.detach().cpu().numpy()
    assert cam.image_shape == (6, 8) and cam.num_rays == 48 and cam._data.image_shape == (6, 8)
    np.testing.assert_allclose(t(cam._data.intrinsic_matrices), G["cls_intrinsics"], **TOL)
    np.testing.assert_allclose(t(cam.ray_directions), G["cls_ray_directions"], **TOL)
    np.testing.assert_allclose(t(cam._offset_pos), G["cls_offset_pos0"], **TOL)
    np.testing.assert_allclose(t(cam._offset_quat), G["cls_offset_quat0"], **TOL)
    # configuration calls
    cam.set_world_poses(dev(G["cls_set_pos"]), dev(G["cls_set_quat"]), env_ids=torch.tensor([0, 1, 2], device=device), convention="ros")
    cam.set_world_poses_from_view(dev(G["cls_eyes"]), dev(G["cls_targets"]), env_ids=torch.tensor([3, 4], device=device))
    np.testing.assert_allclose(t(cam._offset_pos), G["cls_offset_pos"], **TOL)
    np.testing.assert_allclose(t(cam._offset_quat), G["cls_offset_quat"], **TOL)
    np.testing.assert_allclose(t(cam._data.pos_w), G["cls_pos_w_cfg"], **TOL)
    np.testing.assert_allclose(t(cam._data.quat_w_world), G["cls_quat_w_cfg"], **TOL)
    cam.set_intrinsic_matrices(dev(G["cls_K2"]), focal_length=2.0, env_ids=torch.tensor([1, 4], device=device))
    np.testing.assert_allclose(t(cam.ray_directions), G["cls_ray_directions2"], **TOL)
    # one update of every env (all flags start set), read through .data
    cam.update(0.005)
    data = cam.data
    assert t(cam.frame).tolist() == [1] * n and not cam._is_outdated.any()
    np.testing.assert_allclose(t(data.pos_w), G["cls_pos_w"], **TOL)
    np.testing.assert_allclose(t(data.quat_w_world), G["cls_quat_w"], **TOL)

    def close(got, exp, what):
        got = t(got)
        assert got.shape == exp.shape, what
        same = (np.isnan(got) == np.isnan(exp)) & (np.isinf(got) == np.isinf(exp))
        # a ray through a triangle edge may land on either side of fp32 rounding of the rotated direction: allow a few
        assert same.mean() > 0.99, f"{what}: inf / nan pattern differs"
        fin = np.isfinite(exp) & np.isfinite(got)
        bad = ~np.isclose(got[fin], exp[fin], rtol=1e-4, atol=1e-4)
        assert bad.mean() < 0.02, f"{what}: {bad.sum()} of {bad.size} values differ"

    close(data.output["distance_to_camera"], G["cls_camera"], "distance_to_camera")
    close(data.output["distance_to_image_plane"], G["cls_image_plane"], "distance_to_image_plane")
    close(data.output["normals"], G["cls_normals"], "normals")
    close(cam.ray_hits_w, G["cls_hits"], "ray_hits_w")
    # a second read does nothing (no env is outdated); after reset + update every env is recomputed
    cam.data
    assert t(cam.frame).tolist() == [1] * n
    cam.reset()
    assert t(cam.frame).tolist() == [0] * n and cam._is_outdated.all()
    cam.update(0.005, force_recompute=True)
    assert t(cam.frame).tolist() == [1] * n
    # a sensor with an update period only recomputes once the period has passed (sensor_base.py:197-205)
    cam.cfg.update_period = 0.02
    cam.update(0.005)
    cam.data
    assert t(cam.frame).tolist() == [1] * n
    for _ in range(4):
        cam.update(0.005)
    cam.data
    assert t(cam.frame).tolist() == [2] * n
    # the quaternion in the other conventions (CameraData.quat_w_ros / quat_w_opengl)
    np.testing.assert_allclose(np.abs((t(data.quat_w_ros) * cm.convert_camera_frame_orientation_convention(G["cls_quat_w"], "world", "ros")).sum(-1)), 1.0, atol=1e-5)
