"""ndtgpu_fuser_update_batch: NDTFeatureFuserHMT::update for a batch of independent fusers as ONE asynchronous call (-m gpu).

Reference: ndt_feature/src/ndt_feature_src/ndt_feature_fuser_hmt.cpp:65-102 (initialize), 108-512 (update).  The bank must
give, slot by slot and update by update,
  * the BITS of the separate calls it replaces -- scan moved into the node map's frame, ndtgpu_mapset_build on the node map's
    lattice, ndtgpu_match_fusion_feat_batch / ndtgpu_match_fusion_batch / ndtgpu_match_batch (3-DoF), ndtgpu_covariance_batch,
    the pose update, ndtgpu_mapset_add_cloud -- driven here with the inputs ndtgpu_fuser_prepare hands out;
  * the CPU oracle's poses (1e-4 m / 1e-4 rad, BASELINE.json) and node maps when the oracle walks the same sequence (its
    ray-traced insert in the order-free form, which is what the HIP path implements: tests/test_gpu_fuse.py)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RES, RANGE = 0.5, 30.0
NODE_SIZE = [100.0, 100.0, 1.0]
DET_FIELDS = ["converged", "iterations", "fevals", "exit_code", "score", "n_source", "n_target", "pair_terms_g", "pair_terms_h"]


@pytest.fixture(scope="module")
def N():
    import ndt_feature_graph_amd as N
    if N.device_count() < 1:
        pytest.fail("no HIP device visible: the HIP path cannot run (there is no CPU fallback)")
    return N


@pytest.fixture(scope="module")
def O():
    import oracle
    return oracle


# ---- the pose arithmetic of the post-registration step, with the rounding of csrc/ndt_pose.h (scalar IEEE doubles, products
# rounded before they are added, the same loops) -- so that poses can be compared bit for bit
def mm(A, B):
    C = np.zeros((4, 4))
    for c in range(4):
        for r in range(4):
            s = 0.0
            for k in range(4):
                s += float(A[r, k]) * float(B[k, c])
            C[r, c] = s
    return C


def rigid_inv(A):
    C = np.eye(4)
    C[:3, :3] = A[:3, :3].T
    for i in range(3):
        s = 0.0
        for j in range(3):
            s += float(C[i, j]) * float(A[j, 3])
        C[i, 3] = -s
    return C


def euler012(T):
    R = T[:3, :3]
    r0 = math.atan2(R[1, 2], R[2, 2])
    c2 = math.sqrt(R[0, 0] * R[0, 0] + R[0, 1] * R[0, 1])
    if r0 > 0.0:
        r0 -= math.pi
        r1 = math.atan2(-R[0, 2], -c2)
    else:
        r1 = math.atan2(-R[0, 2], c2)
    s1, c1 = math.sin(r0), math.cos(r0)
    r2 = math.atan2(s1 * R[2, 0] - c1 * R[1, 0], c1 * R[1, 1] - s1 * R[2, 1])
    return np.array([-r0, -r1, -r2])


def move_cloud(T, xyz):
    """lslgeneric::transformPointCloudInPlace: float points through a double matrix, products rounded, result rounded to float"""
    x, y, z = (xyz[..., k].astype(np.float64) for k in range(3))
    out = np.empty(xyz.shape[:-1] + (3,), dtype=np.float32)
    for r in range(3):
        out[..., r] = (T[r, 0] * x + T[r, 1] * y + T[r, 2] * z + T[r, 3]).astype(np.float32)
    return out


def pose2d(x, y, yaw):
    T = np.eye(4)
    c, s = math.cos(yaw), math.sin(yaw)
    T[:2, :2] = [[c, -s], [s, c]]
    T[0, 3], T[1, 3] = x, y
    return T


def trajectory(n_fusers, n_steps, seed0=4100):
    """per fuser a room (seed) and a short path: true poses, the scans taken there (sensor frame), noisy odometry increments"""
    import torch
    from ndt_feature_graph_amd import synth
    rng = np.random.default_rng(7)
    seeds = np.arange(seed0, seed0 + n_fusers)
    poses = np.zeros((n_steps + 1, n_fusers, 3))
    poses[0, :, :2] = rng.uniform(-1.0, 1.0, size=(n_fusers, 2))
    poses[0, :, 2] = rng.uniform(-0.3, 0.3, size=n_fusers)
    for s in range(1, n_steps + 1):
        step = np.stack([rng.uniform(0.15, 0.35, n_fusers), rng.uniform(-0.08, 0.08, n_fusers), rng.uniform(-0.06, 0.06, n_fusers)], axis=1)
        for k in range(n_fusers):
            c, sn = math.cos(poses[s - 1, k, 2]), math.sin(poses[s - 1, k, 2])
            poses[s, k, 0] = poses[s - 1, k, 0] + c * step[k, 0] - sn * step[k, 1]
            poses[s, k, 1] = poses[s - 1, k, 1] + sn * step[k, 0] + c * step[k, 1]
            poses[s, k, 2] = poses[s - 1, k, 2] + step[k, 2]
    scans = [synth.scan_2d(torch.as_tensor(seeds), torch.as_tensor(poses[s]), NPTS, noise_stream=s).numpy() for s in range(n_steps + 1)]
    Tm = np.zeros((n_steps, n_fusers, 4, 4))
    for s in range(n_steps):
        for k in range(n_fusers):
            true = np.linalg.inv(pose2d(*poses[s, k])) @ pose2d(*poses[s + 1, k])
            noise = pose2d(rng.normal(0, 0.03), rng.normal(0, 0.02), rng.normal(0, 0.01))
            Tm[s, k] = true @ noise
    return poses, scans, Tm


NPTS = 20000


def feat_of(pp):
    """the 40 odometry cell pairs of fuser_hmt.cpp:322-339 from what ndtgpu_fuser_prepare hands out"""
    sm = np.tile(pp["feat_src_mean"], (40, 1))
    tm = np.tile(pp["feat_tgt_mean"], (40, 1))
    sc = np.tile(pp["feat_cov_rotated"], (40, 1))
    sc[39] = pp["feat_cov_plain"]
    tc = np.tile(pp["feat_cov_rotated"], (40, 1))
    return sm, sc, tm, tc


def cells_bits_equal(a, b, what):
    for x, y, name in zip(a, b, ("mean", "cov", "idx", "n")):
        assert np.array_equal(x, y), "%s: %s differs" % (what, name)


@pytest.mark.parametrize("mode", ["default", "no_odom_cells", "fusion2d", "plain_d2d"])
def test_bank_equals_the_calls_it_replaces(N, mode):
    """three updates of a few fusers: after every update the registered increment, the result fields, the pose, the scan map
    and the node map are the bits of the step-by-step path"""
    import torch
    from ndt_feature_graph_amd import binding
    dev = torch.device("cuda", 0)
    Bn = {"default": 4, "no_odom_cells": 12, "fusion2d": 12, "plain_d2d": 12}[mode]      # (> 8 pairs: both paths on the persistent matcher)
    fields = dict(resolution=RES, map_size_x=NODE_SIZE[0], map_size_y=NODE_SIZE[1], map_size_z=NODE_SIZE[2], sensor_range=RANGE,
                  neighbours=2, delta_score=1e-6, max_cells=4096,
                  sensor_pose=pose2d(0.25, -0.05, 0.02))
    if mode == "no_odom_cells":
        fields.update(use_odom=0)
    if mode == "fusion2d":
        fields.update(fusion2d=1)
    if mode == "plain_d2d":
        fields.update(use_odom=0, use_soft_constraints=0, use_tikhonov=0, check_consistency=1, max_translation_norm=0.04)
    prm = N.fuser_params(**fields)
    sensor = pose2d(0.25, -0.05, 0.02)
    n_steps = 3
    poses, scans, Tm = trajectory(Bn, n_steps)
    bank = N.FuserBank(prm, Bn)
    nodes, scan_maps = bank.mapsets()
    # ---- the step-by-step path: its own node maps and scan maps
    ref_nodes = N.MapSet(RES, [0, 0, 0], NODE_SIZE, n_maps=Bn, max_cells=4096)
    ref_nodes.enable_occupancy()
    local = RANGE + 3 * RES
    ref_scans = N.MapSet(RES, [0, 0, 0], [local, local, NODE_SIZE[2]], n_maps=Bn, max_cells=4096)
    Tnow = [init_pose(k) for k in range(Bn)]
    Tlast = [t.copy() for t in Tnow]
    # initialize (fuser_hmt.cpp:65-102)
    cl0 = torch.as_tensor(scans[0], device=dev).contiguous()
    bank.initialize(np.stack(Tnow), cl0)
    moved = np.stack([move_cloud(Tnow[k], move_cloud(sensor, scans[0][k])) for k in range(Bn)])
    for k in range(Bn):
        ref_nodes.set_centre(k, [Tnow[k][0, 3], Tnow[k][1, 3], 0.0])
    ref_nodes.add_cloud(torch.as_tensor(moved, device=dev).contiguous(), np.stack([mm(Tnow[k], sensor)[:3, 3] for k in range(Bn)]),
                        maxz=100.0, sensor_noise=0.1)
    torch.cuda.synchronize()
    T_b, _ = bank.poses()
    for k in range(Bn):
        assert np.array_equal(T_b[k], Tnow[k])
        cells_bits_equal(nodes.export_cells(k), ref_nodes.export_cells(k), "node map %d after initialize" % k)
        assert np.array_equal(nodes.occupancy(k), ref_nodes.occupancy(k))
    failures = 0
    cov_acc = [np.zeros((3, 3)) for _ in range(Bn)]
    for s in range(n_steps):
        cl = torch.as_tensor(scans[s + 1], device=dev).contiguous()
        bank.update(Tm[s], cl)
        T_b, r_b = bank.poses()
        # -- the separate calls
        pps = [N.fuser_prepare(prm, Tnow[k], Tm[s, k], node_centres(Bn)[k]) for k in range(Bn)]
        in_node = np.stack([move_cloud(np.array(pps[k]["Tscan"]).reshape(4, 4).T, scans[s + 1][k]) for k in range(Bn)])
        for k in range(Bn):
            ref_scans.set_centre(k, pps[k]["scan_centre"])
        ref_scans.build(torch.as_tensor(in_node, device=dev).contiguous(), range_limit=RANGE,
                        range_origins=np.stack([pps[k]["range_origin"] for k in range(Bn)]))
        idx = np.arange(Bn)
        Tcov = np.stack([pps[k]["Tcov"].reshape(6, 6) for k in range(Bn)])
        common = dict(n_neighbours=2, itr_max=30, delta_score=1e-6, step_control=1)
        if mode == "default":
            T_est, r = binding.match_fusion_feat_batch(ref_nodes, idx, ref_scans, idx, Tm[s], Tcov, [feat_of(pp) for pp in pps],
                                                       use_soft_constraints=True, tikhonov=True, step_control_fusion=True, **common)
        elif mode == "no_odom_cells":
            T_est, r = binding.match_fusion_batch(ref_nodes, idx, ref_scans, idx, Tm[s], Tcov, use_soft_constraints=True, tikhonov=True, **common)
        elif mode == "fusion2d":
            T_est, r = binding.match_batch(ref_nodes, idx, ref_scans, idx, Tm[s], dof_mask=0x23, **common)
        else:
            T_est, r = binding.match_batch(ref_nodes, idx, ref_scans, idx, Tm[s], **common)
        cov6, sing = binding.covariance(ref_nodes, idx, ref_scans, idx, T_est, mode=0, n_neighbours=2)
        new_T, spose = [], []
        for k in range(Bn):
            ok = bool(r["converged"][k])
            if ok:
                diff = mm(rigid_inv(T_est[k]), Tm[s, k])
                gate = (np.linalg.norm(diff[:3, 3]) > prm.max_translation_norm or np.linalg.norm(euler012(diff)) > prm.max_rotation_norm) \
                    and prm.check_consistency
                Tn = mm(Tnow[k], Tm[s, k]) if gate else mm(Tnow[k], T_est[k])
                failures += int(gate)
                assert r_b["registration_failure"][k] == int(gate)
            else:
                Tn = mm(Tnow[k], Tm[s, k])
            assert r_b["match_ok"][k] == int(ok)
            if ok and prm.compute_cov:           # fuser_hmt.cpp:399-413: the diagonal of the matcher's covariance accumulates
                cov_acc[k] = cov_acc[k] + np.diag([cov6[k][0, 0], cov6[k][1, 1], cov6[k][5, 5]])
                assert np.allclose(r_b["posecov_mean"][k], [T_est[k][0, 3], T_est[k][1, 3], euler012(T_est[k])[2]], rtol=0, atol=1e-12)
            assert np.array_equal(r_b["posecov"][k].reshape(3, 3).T, cov_acc[k]), (mode, s, k)
            new_T.append(Tn)
            spose.append(mm(Tn, sensor))
        fused = np.stack([move_cloud(spose[k], scans[s + 1][k]) for k in range(Bn)])
        ref_nodes.add_cloud(torch.as_tensor(fused, device=dev).contiguous(), np.stack([sp[:3, 3] for sp in spose]), maxz=25.0, sensor_noise=0.06)
        torch.cuda.synchronize()
        # -- bit for bit
        for k in range(Bn):
            what = "%s, update %d, fuser %d" % (mode, s, k)
            assert np.array_equal(r_b["Tmotion_est"][k].reshape(4, 4).T, T_est[k]), what
            for f in DET_FIELDS:
                assert r_b["match"][f][k] == r[f][k], (what, f)
            assert np.array_equal(T_b[k], new_T[k]), what
            assert np.array_equal(r_b["spose"][k].reshape(4, 4).T, spose[k]), what
            cells_bits_equal(scan_maps.export_cells(k), ref_scans.export_cells(k), "scan map, " + what)
            cells_bits_equal(nodes.export_cells(k), ref_nodes.export_cells(k), "node map, " + what)
            assert np.array_equal(nodes.occupancy(k), ref_nodes.occupancy(k)), what
            if r["converged"][k] and prm.compute_cov:
                assert r_b["cov_singular"][k] == sing[k]
        Tnow = new_T
    if mode == "plain_d2d":
        assert failures > 0, "the consistency gate was meant to fire in this mode (0.04 m against 3 cm odometry noise)"
    print("%s: converged %.2f, iterations %s, exit codes %s" % (mode, r["converged"].mean(), r["iterations"].tolist(), r["exit_code"].tolist()))
    if mode in ("fusion2d", "plain_d2d"):
        assert r["converged"].mean() > 0.5
    bank.close()


def init_pose(k):
    return pose2d(0.02 * k, -0.03 * k, 0.01 * k)


def node_centres(Bn):
    """the node maps' centres: (x, y, 0) of the initial poses (NDTMap::initialize, fuser_hmt.cpp:89)"""
    return [[init_pose(k)[0, 3], init_pose(k)[1, 3], 0.0] for k in range(Bn)]


def test_bank_against_the_oracle(N, O):
    """the oracle walks the same sequence -- ray-traced node map, scan map on its lattice, matchFusion with the soft constraint,
    the Tikhonov term and the 40 odometry cells, pose update, fuse-in: at every update the bank's pose is within 1e-4 m / 1e-4 rad
    of the pose the oracle registers from the same maps, and -- fused in at the bank's pose, so that both sides ray-trace the
    same float points -- its node map holds the oracle's cells"""
    import torch
    dev = torch.device("cuda", 0)
    Bn, n_steps = 3, 3
    sensor = pose2d(0.25, -0.05, 0.02)
    prm = N.fuser_params(resolution=RES, map_size_x=NODE_SIZE[0], map_size_y=NODE_SIZE[1], map_size_z=NODE_SIZE[2], sensor_range=RANGE,
                         neighbours=2, delta_score=1e-6, max_cells=4096, sensor_pose=sensor)
    poses, scans, Tm = trajectory(Bn, n_steps, seed0=5200)
    bank = N.FuserBank(prm, Bn)
    nodes, _ = bank.mapsets()
    Tnow = [init_pose(k) for k in range(Bn)]
    bank.initialize(np.stack(Tnow), torch.as_tensor(scans[0], device=dev).contiguous())
    omaps = []
    for k in range(Bn):
        om = O.OracleMap(RES, [Tnow[k][0, 3], Tnow[k][1, 3], 0.0], NODE_SIZE)
        om.add_point_cloud((Tnow[k] @ sensor)[:3, 3], move_cloud(Tnow[k], move_cloud(sensor, scans[0][k])), maxz=100.0, sensor_noise=0.1, order_free=True)
        om.compute_cells_full()
        omaps.append(om)
    local = RANGE + 3 * RES
    worst = [0.0, 0.0]
    for s in range(n_steps):
        bank.update(Tm[s], torch.as_tensor(scans[s + 1], device=dev).contiguous())
        T_b, r_b = bank.poses()
        for k in range(Bn):
            pp = N.fuser_prepare(prm, Tnow[k], Tm[s, k], node_centres(Bn)[k])
            # (the host step against plain NumPy)
            Tscan = Tnow[k] @ sensor
            assert np.allclose(pp["Tscan"].reshape(4, 4).T, Tscan, rtol=0, atol=1e-14)
            assert np.allclose(pp["scan_centre"], np.array(node_centres(Bn)[k]) + np.floor((Tscan[:3, 3] - node_centres(Bn)[k]) / RES) * RES)
            rel = [Tm[s, k][0, 3], Tm[s, k][1, 3], math.atan2(Tm[s, k][1, 0], Tm[s, k][0, 0])]
            d2, r2 = rel[0] ** 2 + rel[1] ** 2, rel[2] ** 2
            assert np.allclose(np.diag(pp["Tcov"].reshape(6, 6)), [0.005 * d2 + 0.005 * r2, 0.001 * d2 + 0.001 * r2, 1, 1, 1, 0.001 * d2 + 0.001 * r2],
                               rtol=1e-9)
            os_ = O.OracleMap(RES, pp["scan_centre"], [local, local, NODE_SIZE[2]])
            os_.load_points(move_cloud(Tscan, scans[s + 1][k]), RANGE, range_origin=pp["range_origin"])
            os_.compute_cells()
            To, ro = O.match_fusion_feat(omaps[k], os_, Tm[s, k], pp["Tcov"].reshape(6, 6), feat_of(pp), use_soft_constraints=True,
                                         tikhonov=True, step_control_fusion=True, n_neighbours=2, itr_max=30, delta_score=1e-6, step_control=1)
            Tn = Tnow[k] @ (To if ro["converged"] else Tm[s, k])
            sp = r_b["spose"][k].reshape(4, 4).T                              # (the bank's: Tnow * sensor_pose)
            assert np.allclose(sp, T_b[k] @ sensor, rtol=0, atol=1e-14)
            omaps[k].add_point_cloud(sp[:3, 3], move_cloud(sp, scans[s + 1][k]), maxz=25.0, sensor_noise=0.06, order_free=True)
            omaps[k].compute_cells_full()
            dt = float(np.linalg.norm(T_b[k][:3, 3] - Tn[:3, 3]))
            dr = float(2.0 * np.arcsin(min(1.0, np.linalg.norm(T_b[k][:3, :3] - Tn[:3, :3]) / (2.0 * np.sqrt(2.0)))))
            worst = [max(worst[0], dt), max(worst[1], dr)]
            assert dt <= 1e-4 and dr <= 1e-4, (s, k, dt, dr)                 # the tolerance BASELINE.json's north_star states
            assert bool(r_b["match"]["converged"][k]) == ro["converged"] and r_b["match"]["iterations"][k] == ro["iterations"], (s, k)
            g, c = nodes.export_cells(k), omaps[k].export_cells()
            assert np.array_equal(g[2], c[2]), "node map %d after update %d: cell sets differ" % (k, s)
            assert np.array_equal(g[3].astype(np.int64), c[3].astype(np.int64)) and np.max(np.abs(g[0] - c[0])) < 1e-9
            Tnow[k] = T_b[k]
    print("fuser bank vs oracle: worst |dt| %.3e m, |dR| %.3e rad over %d updates" % (worst[0], worst[1], Bn * n_steps))
    bank.close()


def test_bank_rejects_and_orders(N):
    """argument checks; update before initialize; slots updated in two ranges; a second call starts from the first one's poses"""
    import torch
    dev = torch.device("cuda", 0)
    prm = N.fuser_params(resolution=RES, map_size_x=NODE_SIZE[0], map_size_y=NODE_SIZE[1], map_size_z=NODE_SIZE[2], sensor_range=RANGE,
                         neighbours=2, delta_score=1e-6, max_cells=4096)
    with pytest.raises(N.NdtGpuError):
        N.FuserBank(N.fuser_params(resolution=-1.0), 2)
    Bn = 4
    poses, scans, Tm = trajectory(Bn, 2, seed0=6300)
    bank = N.FuserBank(prm, Bn)
    cl = [torch.as_tensor(sc, device=dev).contiguous() for sc in scans]
    with pytest.raises(N.NdtGpuError):
        bank.update(Tm[0], cl[1])                                   # "NDT-FuserHMT: Call Initialize first!!"
    T0 = np.stack([np.eye(4)] * Bn)
    bank.initialize(T0, cl[0])
    # all four at once ...
    bank.update(Tm[0], cl[1]); bank.update(Tm[1], cl[2])
    Ta, ra = bank.poses()
    # ... and in two ranges, two calls each, on a second bank
    bank2 = N.FuserBank(prm, Bn)
    bank2.initialize(T0[:2], cl[0][:2], first=0); bank2.initialize(T0[2:], cl[0][2:], first=2)
    for s in range(2):
        bank2.update(Tm[s][2:], cl[s + 1][2:], first=2)
        bank2.update(Tm[s][:2], cl[s + 1][:2], first=0)
    Tb, _ = bank2.poses()
    assert np.array_equal(Ta, Tb)
    na, _ = bank.mapsets(); nb, _ = bank2.mapsets()
    for k in range(Bn):
        cells_bits_equal(na.export_cells(k), nb.export_cells(k), "node map %d" % k)
    # update_ndt_map = False leaves the node map alone (graph.cpp:73: the step that opens a new node)
    before = na.export_cells(1)
    bank.update(Tm[0], cl[1], update_ndt_map=False)
    bank.poses()
    cells_bits_equal(na.export_cells(1), before, "node map with updateNDTMap = false")
    bank.close(); bank2.close()
