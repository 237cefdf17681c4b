"""ndtgpu_fuser_prepare -- the host step of NDTFeatureFuserHMT::update (ndt_feature_fuser_hmt.cpp:124-146, 166-202, 291-339) -- is a
pure function of the C-ABI and needs no device: checked here against plain NumPy restatements of the reference's formulas
(MotionModel2d::getMeasurementCov, motion_model.cpp:190-207; getCovMatrix6; loadPointCloudCentroid's snapped centre;
pseudoTransformNDTMap of the odometry cells)."""
import math

import numpy as np
import pytest


def pose(x, y, z, roll, pitch, yaw):
    cx, sx, cy, sy, cz, sz = math.cos(roll), math.sin(roll), math.cos(pitch), math.sin(pitch), math.cos(yaw), math.sin(yaw)
    T = np.eye(4)
    T[:3, :3] = [[cy * cz, -cy * sz, sy], [cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy], [sx * sz - cx * sy * cz, sx * cz + cx * sy * sz, cx * cy]]
    T[:3, 3] = [x, y, z]
    return T


@pytest.mark.parametrize("seed", range(6))
def test_prepare_against_numpy(seed):
    import ndt_feature_graph_amd as N
    rng = np.random.default_rng(seed)
    sensor = pose(0.3, -0.1, 0.05, 0.0, 0.0, rng.uniform(-0.2, 0.2))
    res = [0.5, 1.0, 0.25][seed % 3]
    prm = N.fuser_params(resolution=res, sensor_range=30.0, sensor_pose=sensor, motion_Dd=0.007, motion_Ct=0.002)
    Tnow = pose(rng.uniform(-5, 5), rng.uniform(-5, 5), 0.0, 0.0, 0.0, rng.uniform(-3, 3))
    Tm = pose(rng.uniform(0.05, 0.6), rng.uniform(-0.1, 0.1), 0.0, 0.0, 0.0, rng.uniform(-0.3, 0.3))
    centre = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), 0.0])
    pp = N.fuser_prepare(prm, Tnow, Tm, centre)
    # the scan frame and the scan map's centre on the node map's lattice
    Tscan = Tnow @ sensor
    assert np.allclose(pp["Tscan"].reshape(4, 4).T, Tscan, rtol=0, atol=1e-14)
    assert np.allclose(pp["range_origin"], Tscan[:3, 3], rtol=0, atol=1e-14)
    want = centre + np.floor((Tscan[:3, 3] - centre) / res) * res
    assert np.allclose(pp["scan_centre"], want, rtol=0, atol=1e-12)
    assert np.all(np.abs((pp["scan_centre"] - centre) / res - np.rint((pp["scan_centre"] - centre) / res)) < 1e-9)      # on the lattice
    assert np.all(pp["scan_centre"] <= Tscan[:3, 3] + 1e-12) and np.all(Tscan[:3, 3] - pp["scan_centre"] < res + 1e-12)
    # the odometry model: Eliazar-style diagonal, the yaw of eulerAngles(0, 1, 2) of a planar motion
    yaw = math.atan2(Tm[1, 0], Tm[0, 0])
    d2, r2 = Tm[0, 3] ** 2 + Tm[1, 3] ** 2, yaw ** 2
    R = [0.007 * d2 + 0.005 * r2, 0.001 * d2 + 0.002 * r2, 0.001 * d2 + 0.001 * r2]
    Tcov = pp["Tcov"].reshape(6, 6)
    assert np.allclose(np.diag(Tcov), [R[0], R[1], 1.0, 1.0, 1.0, R[2]], rtol=1e-9, atol=0)
    assert np.count_nonzero(Tcov - np.diag(np.diag(Tcov))) == 0
    oc = pp["odom_cov"].reshape(3, 3)
    assert np.allclose(np.diag(oc), [R[0], R[1], 0.01], rtol=1e-9) and np.count_nonzero(oc - np.diag(np.diag(oc))) == 0
    # the odometry cell pair, both moved into the node map's frame by Tnow
    assert np.allclose(pp["feat_src_mean"], Tnow[:3, 3], rtol=0, atol=1e-14)
    assert np.allclose(pp["feat_tgt_mean"], (Tnow @ np.append(Tm[:3, 3], 1.0))[:3], rtol=0, atol=1e-13)
    rot = Tnow[:3, :3] @ oc @ Tnow[:3, :3].T
    ij = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    assert np.allclose(pp["feat_cov_rotated"], [rot[a, b] for a, b in ij], rtol=1e-12, atol=1e-18)
    assert np.allclose(pp["feat_cov_plain"], [oc[a, b] for a, b in ij], rtol=0, atol=0)


def test_prepare_rejects_and_defaults():
    import ndt_feature_graph_amd as N
    p = N.fuser_params()
    # NDTFeatureFuserHMT::Params() (ndt_feature_fuser_hmt.h:58-101) and MotionModel2d::Params() (motion_model.hpp:128-136)
    assert (p.resolution, p.map_size_x, p.map_size_y, p.map_size_z, p.sensor_range) == (1.0, 40.0, 40.0, 10.0, 3.0)
    assert (p.use_odom, p.neighbours, p.stepcontrol, p.itr_max, p.use_soft_constraints, p.compute_cov, p.step_control_fusion, p.use_tikhonov) == (1, 0, 1, 30, 1, 1, 1, 1)
    assert (p.check_consistency, p.fuse_incomplete, p.force_odom_as_est, p.fusion2d, p.all_matches_valid, p.discard_cells) == (0, 0, 0, 0, 0, 0)
    assert abs(p.delta_score - 10e-4) < 1e-18 and abs(p.max_rotation_norm - math.pi / 4) < 1e-15 and p.max_translation_norm == 1.0
    assert (p.motion_Cd, p.motion_Ct, p.motion_Dd, p.motion_Dt, p.motion_Td, p.motion_Tt) == (0.001, 0.001, 0.005, 0.005, 0.001, 0.001)
    assert list(p.sensor_pose) == [1.0 if k % 5 == 0 else 0.0 for k in range(16)]
    with pytest.raises(N.NdtGpuError):
        N.fuser_prepare(N.fuser_params(resolution=0.0), np.eye(4), np.eye(4), [0, 0, 0])
