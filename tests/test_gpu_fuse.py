"""Incremental (fused) node maps, occupancy and the overlap score on the GPU vs the CPU oracle (-m gpu).

Reference path: NDTMap::initialize + addPointCloud + computeNDTCells(SAMPLE_VARIANCE, 1e5, 255, ..) as the fuser calls
them (ndt_feature_fuser_hmt.cpp:87-94, 482-487) -- the maps NDTFeatureGraph actually registers (graph.cpp:273) --
and ndt_feature::overlapNDTOccupancyScore (ndt_feature_node.h:213-252).

Bars: the set of Gaussian cells and their point counts N bit-exact; means 1e-9 m, covariances 1e-8 relative;
occupancy log-odds 1e-4 (a float, sums of ~1e3 float updates; the HIP path adds them exactly, see below); the
integer nb_sum of the overlap score exact, the score 1e-12.

The HIP path implements the ORDER-FREE semantics of the ray-traced insert (oracle `order_free=True`): every beam
sees the cells as they were when the call started and a cell's updates are summed exactly.  The reference walks the
beams one after the other; `test_order_free_vs_reference_order` measures what that changes on the fused maps."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LOG15 = float(np.log(0.6 / (1.0 - 0.6)))


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


def rot_angle(Ra, Rb):
    return float(2.0 * np.arcsin(min(1.0, np.linalg.norm(Ra - Rb) / (2.0 * np.sqrt(2.0)))))


def cells_equal(gpu, cpu, res, cov_rel=1e-8):
    gm, gc, gi, gn = gpu
    cm, cc, ci, cn = cpu
    assert len(gn) == len(cn), "number of Gaussian cells differs: %d vs %d" % (len(gn), len(cn))
    assert np.array_equal(gi, ci), "cell index sets differ"
    assert np.array_equal(gn.astype(np.int64), cn.astype(np.int64)), "per-cell N differs"
    if len(gn) == 0:
        return
    assert np.max(np.abs(gm - cm)) < 1e-9 * max(1.0, res)
    scale = np.max(np.abs(cc), axis=(1, 2), keepdims=True)
    assert np.max(np.abs(gc - cc) / scale) < cov_rel


def node_scans(seed, n, n_pts, step=0.15):
    """n scans of one room from a short trajectory, transformed into the frame of the first pose (the node frame):
    (world points [n, N, 3] float32 with NaN where the beam found nothing, sensor origins [n, 3], poses [n,4,4])."""
    from ndt_feature_graph_amd import synth
    poses = np.array([[step * k, 0.05 * np.sin(k), 0.02 * k] for k in range(n)])
    scans = synth.scan_2d([seed] * n, poses, n_pts).numpy()
    T = synth.pose2d_to_T(poses).numpy()
    T = np.linalg.inv(T[0]) @ T
    out = np.empty_like(scans)
    for k in range(n):
        out[k] = (scans[k].astype(np.float64) @ T[k][:3, :3].T + T[k][:3, 3]).astype(np.float32)
    return out, T[:, :3, 3].copy(), T


def fuse_both(N, O, clouds, origins, res, size, order_free=True, check_every=True, **prm):
    """initialize + (addPointCloud, computeNDTCells) per cloud on the GPU and in the oracle; first cloud with the
    arguments of NDTFeatureFuserHMT::initialize (0.1, 100, 0.1), the others with those of update (0.06, 25)."""
    ms = N.MapSet(res, [0, 0, 0], size, n_maps=1, max_cells=prm.pop("max_cells", 0))
    ms.enable_occupancy()
    om = O.OracleMap(res, [0, 0, 0], size)
    for k in range(len(clouds)):
        kw = dict(maxz=100.0, sensor_noise=0.1) if k == 0 else dict(maxz=25.0, sensor_noise=0.06)
        kw.update(prm)
        ms.add_cloud(clouds[k][None], origins[k][None], **kw)
        om.add_point_cloud(origins[k], clouds[k], maxz=kw["maxz"], sensor_noise=kw["sensor_noise"],
                           occupancy_limit=kw.get("occupancy_limit", 255.0), order_free=order_free)
        om.compute_cells_full(maxnumpoints=kw.get("maxnumpoints", 1e5), occupancy_limit=kw.get("occupancy_limit", 255.0))
        if check_every or k == len(clouds) - 1:
            cells_equal(ms.export_cells(0), om.export_cells(), res)
            og, oo = ms.occupancy(0), om.occupancy()
            assert np.max(np.abs(og - oo)) < 1e-4, (k, float(np.max(np.abs(og - oo))))
    return ms, om


def test_fuse_parity_2d_node_map(N, O):
    """A node map as the reference builds it: 16 scans of 20 k points fused one after the other."""
    clouds, origins, _ = node_scans(7, 16, 20000)
    ms, om = fuse_both(N, O, clouds, origins, 0.5, [100, 100, 1])
    mean, cov, idx, n = ms.export_cells(0)
    assert len(n) > 100 and n.max() > 20000 // 100          # cells hold the points of many scans
    occ = ms.occupancy(0)
    assert occ.min() < -1.0 and occ.max() > 50.0              # free space was carved, walls are firmly occupied
    assert ms.counters(0)["overflow"] == 0


def test_fuse_parity_full_size_scans_40(N, O):
    """40 scans x 100 k points into one 0.5 m node map (the upper end of what a node holds); N saturates nowhere yet."""
    clouds, origins, _ = node_scans(11, 40, 100000, step=0.05)
    ms, om = fuse_both(N, O, clouds, origins, 0.5, [100, 100, 1], check_every=False)
    assert ms.export_cells(0)[3].max() > 30000


def test_fuse_saturation_and_limits(N, O):
    """maxnumpoints small enough that N saturates ("sliding average"), a tight occupancy limit, n_min = 6."""
    clouds, origins, _ = node_scans(3, 8, 20000)
    ms, om = fuse_both(N, O, clouds, origins, 0.5, [100, 100, 1], maxnumpoints=300.0, occupancy_limit=20.0)
    n = ms.export_cells(0)[3]
    assert n.max() == 300 and (n == 300).sum() > 10
    occ = ms.occupancy(0)
    assert occ.max() == 20.0 and occ.min() >= -20.0


def test_fuse_batch_of_maps_equals_single_maps(N):
    """B maps updated in one call = B single-map calls (bit for bit), and a second cloud per map lands in the right map."""
    B = 6
    data = [node_scans(20 + b, 3, 8000) for b in range(B)]
    one = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=B)
    one.enable_occupancy()
    for k in range(3):
        one.add_cloud(np.stack([data[b][0][k] for b in range(B)]), np.stack([data[b][1][k] for b in range(B)]))
    for b in range(B):
        single = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=1)
        single.enable_occupancy()
        for k in range(3):
            single.add_cloud(data[b][0][k][None], data[b][1][k][None])
        for x, y in zip(one.export_cells(b), single.export_cells(0)):
            assert np.array_equal(x, y)
        assert np.array_equal(one.occupancy(b), single.occupancy(0))


def test_merge_equals_batch_covariance_numpy(N):
    """Golden, independent of the oracle: Chan's pairwise update of (N, mean, S) is algebraically the sample covariance
    of ALL points; three batches into one cell (sensor inside the cell: no beam crosses anything) vs np.cov."""
    g = np.random.default_rng(5)
    centre = np.array([3.0, -2.0, 0.0])           # a cell centre of the 1 m grid
    A = g.normal(size=(3, 3)) * 0.08
    batches = [(centre + g.normal(size=(n, 3)) @ A.T).astype(np.float32) for n in (40, 7, 300)]
    assert all(np.all(np.abs(b - centre) < 0.45) for b in batches)
    ms = N.MapSet(1.0, [0, 0, 0], [10, 10, 2])
    ms.enable_occupancy()
    seen = []
    for b in batches:
        ms.add_cloud(b[None], centre[None])
        seen.append(b.astype(np.float64))
        allp = np.concatenate(seen)
        mean, cov, idx, n = ms.export_cells(0)
        assert len(n) == 1 and n[0] == len(allp)
        np.testing.assert_allclose(mean[0], allp.mean(axis=0), rtol=0, atol=1e-12)
        want = np.cov(allp.T, ddof=1)
        ev = np.linalg.eigvalsh(want)
        assert ev[0] > ev[2] / 1000.0                      # no eigenvalue floor: the covariance comes through unchanged
        assert np.max(np.abs(cov[0] - want)) < 1e-10 * np.max(np.abs(want))
    occ = ms.occupancy(0)
    k = tuple(idx[0])
    assert occ[k] == np.float32(np.float32(np.float32(40 * LOG15) + np.float32(7 * LOG15)) + np.float32(300 * LOG15))
    assert np.count_nonzero(occ) == 1
    # saturation, against the recurrence written out in NumPy
    ms2 = N.MapSet(1.0, [0, 0, 0], [10, 10, 2])
    ms2.enable_occupancy()
    Nn, msum, S = 0.0, None, None
    for b in batches:
        ms2.add_cloud(b[None], centre[None], maxnumpoints=100.0)
        p = b.astype(np.float64)
        if Nn == 0:
            Nn, msum = float(len(p)), p.sum(axis=0)
            S = (p - p.mean(axis=0)).T @ (p - p.mean(axis=0))
        else:
            n, T2 = float(len(p)), p.sum(axis=0)
            c2 = (p - p.mean(axis=0)).T @ (p - p.mean(axis=0))
            c3 = msum * (n / Nn) - T2
            S = S + c2 + (Nn / (n * (Nn + n))) * np.outer(c3, c3)
            msum = msum + T2
            Nn += n
            if Nn > 100.0:
                msum, S, Nn = msum * (100.0 / Nn), S * (99.0 / (Nn - 1.0)), 100.0
        mean, cov, idx, n = ms2.export_cells(0)
        assert n[0] == int(Nn)
        np.testing.assert_allclose(mean[0], msum / Nn, rtol=0, atol=1e-12)
        want = S / (Nn - 1.0)
        assert np.max(np.abs(cov[0] - want)) < 1e-10 * np.max(np.abs(want))


def _trace_numpy(origin, end, res, centre, size):
    """LazyGrid::traceLine restated in NumPy (independent of the oracle): sampled walk, float samples."""
    diff = end.astype(np.float64) - origin
    l = np.linalg.norm(diff)
    n = int(l / res)
    slots, last = [], (0, 0, 0)
    for i in range(n - 2):
        pt = (origin + np.float64(np.float32(i + 1)) * (diff / np.float64(np.float32(n)))).astype(np.float32)
        idx = tuple(int(np.floor((np.float64(pt[a]) - centre[a]) / res + 0.5) + size[a] / 2.0) for a in range(3))
        if idx == last:
            continue
        last = idx
        if all(0 <= idx[a] < size[a] for a in range(3)):
            slots.append(idx)
    return slots


def test_ray_walk_known_answer(N, O):
    """A few beams through an empty grid: exactly the cells of the sampled walk get -0.2 per crossing beam, the hit cell
    gets its point; a beam whose end is above maxz does nothing at all; NaN and out-of-range points are ignored."""
    res, size_m, size = 0.5, [20.0, 20.0, 2.0], (40, 40, 4)
    origin = np.array([0.3, -0.2, 0.1])
    ends = np.array([[7.3, 2.1, 0.2], [-6.2, 5.5, 0.4], [0.9, -8.8, 0.3], [5.0, 5.0, 0.35], [5.0, 5.0, 0.35],
                     [3.0, 3.0, 0.9], [np.nan, 1.0, 0.0], [150.0, 160.0, 0.0], [0.6, 0.1, 0.1]], dtype=np.float32)
    ms = N.MapSet(res, [0, 0, 0], size_m)
    ms.enable_occupancy()
    ms.add_cloud(ends[None], origin[None], maxz=0.8, sensor_noise=0.1)
    want = np.zeros(size, dtype=np.float64)
    for e in ends:
        if np.isnan(e).any() or np.linalg.norm(e.astype(np.float64) - origin) > 200.0 or e[2] > 0.8:
            continue
        for idx in _trace_numpy(origin, e, res, [0, 0, 0], size):
            want[idx] += np.float64(np.float32(-0.2))
        hit = tuple(int(np.floor(np.float64(e[a]) / res + 0.5) + size[a] / 2.0) for a in range(3))
        if all(0 <= hit[a] < size[a] for a in range(3)):
            want[hit] += LOG15
    occ = ms.occupancy(0)
    assert np.count_nonzero(want) > 30
    assert np.array_equal(occ != 0, want != 0)
    assert np.max(np.abs(occ - want.astype(np.float32))) < 1e-6
    assert ms.num_cells(0) == 0                                   # nowhere three points in a cell
    om = O.OracleMap(res, [0, 0, 0], size_m)
    om.add_point_cloud(origin, ends, maxz=0.8, sensor_noise=0.1, order_free=True)
    om.compute_cells_full()
    assert np.max(np.abs(om.occupancy() - occ)) < 1e-6


@pytest.mark.parametrize("seeds", [[7, 2, 10], [12, 14, 21]])
def test_order_free_vs_reference_order(N, O, seeds):
    """What the order-free semantics change against the reference's beam-after-beam walk (a cell that loses its Gaussian
    half way through a cloud is treated as empty by the remaining beams; float accumulation of the occupancy): on
    16-scan node maps NOTHING that the matcher sees -- exactly the same Gaussian cells, the same N per cell, the same
    moments; occupancies within float accumulation error.  (tests/test_oracle.py runs the same comparison, oracle
    against oracle, on 20 node maps.)"""
    for seed in seeds:
        clouds, origins, _ = node_scans(seed, 16, 20000)
        ms, om_free = fuse_both(N, O, clouds, origins, 0.5, [100, 100, 1], check_every=False)
        om_seq = O.OracleMap(0.5, [0, 0, 0], [100, 100, 1])
        for k in range(len(clouds)):
            kw = dict(maxz=100.0, sensor_noise=0.1) if k == 0 else dict(maxz=25.0, sensor_noise=0.06)
            om_seq.add_point_cloud(origins[k], clouds[k], order_free=False, **kw)
            om_seq.compute_cells_full()
        cells_equal(ms.export_cells(0), om_seq.export_cells(), 0.5)       # index sets, N: exact; moments 1e-9 / 1e-8
        do = np.abs(ms.occupancy(0) - om_seq.occupancy())
        assert do.max() < 1e-2, (seed, float(do.max()))


def test_plain_build_leaves_occupancy(N, O):
    """loadPointCloud + computeNDTCells on an occupancy-enabled set: occ = min(n log 1.5, 255) in every touched cell."""
    from ndt_feature_graph_amd import synth
    pts = synth.pair_2d([4], 30000)["fixed"].numpy()
    ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=1)
    ms.enable_occupancy()
    ms.build(pts, range_limit=30.0)
    om = O.OracleMap(0.5, [0, 0, 0], [100, 100, 1])
    om.load_points(pts[0], 30.0)
    om.compute_cells()
    assert np.array_equal(ms.occupancy(0), om.occupancy())
    assert ms.occupancy(0).max() > 100.0
    ms.build(pts[:, :3000], range_limit=30.0, first=0)                     # a rebuild starts from a fresh map
    om.load_points(pts[0][:3000], 30.0)
    om.compute_cells()
    assert np.array_equal(ms.occupancy(0), om.occupancy())


def test_overlap_score_parity(N, O, monkeypatch):
    """overlapNDTOccupancyScore over all ordered pairs of five fused node maps, at the true relative poses and at
    perturbed ones: nb_sum exact, score 1e-12; identical maps at the identity score 0; disjoint maps score 1."""
    from ndt_feature_graph_amd import synth
    n_nodes, per_node = 5, 4
    poses = np.array([[0.6 * k, 0.1 * np.sin(k), 0.03 * k] for k in range(n_nodes * per_node)])
    scans = synth.scan_2d([9] * len(poses), poses, 15000).numpy()
    Tw = synth.pose2d_to_T(poses).numpy()
    ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=n_nodes)
    ms.enable_occupancy()
    oms = [O.OracleMap(0.5, [0, 0, 0], [100, 100, 1]) for _ in range(n_nodes)]
    for j in range(per_node):
        cl, org = [], []
        for nd in range(n_nodes):
            k = nd * per_node + j
            Tl = np.linalg.inv(Tw[nd * per_node]) @ Tw[k]
            cl.append((scans[k].astype(np.float64) @ Tl[:3, :3].T + Tl[:3, 3]).astype(np.float32))
            org.append(Tl[:3, 3])
            oms[nd].add_point_cloud(org[-1], cl[-1], maxz=25.0, sensor_noise=0.06, order_free=True)
            oms[nd].compute_cells_full()
        ms.add_cloud(np.stack(cl), np.stack(org))
    for nd in range(n_nodes):
        assert np.max(np.abs(ms.occupancy(nd) - oms[nd].occupancy())) < 1e-4
    g = np.random.default_rng(1)
    ref, mov, Ts = [], [], []
    for a in range(n_nodes):
        for b in range(n_nodes):
            T = np.linalg.inv(Tw[a * per_node]) @ Tw[b * per_node]
            for pert in (0, 1):
                P = np.eye(4)
                if pert:
                    P = synth.pose2d_to_T(g.normal(scale=[0.2, 0.2, 0.05])[None]).numpy()[0]
                ref.append(a); mov.append(b); Ts.append(T @ P)
    Ts = np.stack(Ts)
    score, nb = N.overlap_score(ms, ref, ms, mov, Ts)
    # the oracle works on ITS occupancies: a float that differs in the last place would move the score by ~1e-8, so
    # the GPU's own occupancies are scored by the oracle formula as well (exact) and the oracle's maps to 1e-6
    for k in range(len(ref)):
        so, nbo = O.overlap_score(oms[ref[k]], oms[mov[k]], Ts[k])
        assert nb[k] == nbo, (k, nb[k], nbo)
        assert abs(score[k] - so) < 1e-6 * max(so, 1e-3), (k, score[k], so)
    # (a batch in which moving maps recur goes through lists of their cells with a reading; the dense kernel on the same
    #  batch: the same counts, the same score up to the order of one sum)
    monkeypatch.setenv("NDTGPU_OVERLAP_DENSE", "1")
    score_d, nb_d = N.overlap_score(ms, ref, ms, mov, Ts)
    monkeypatch.delenv("NDTGPU_OVERLAP_DENSE")
    assert np.array_equal(nb, nb_d) and np.max(np.abs(score - score_d)) <= 1e-13 * np.max(np.abs(score_d))
    score_r, nb_r = N.overlap_score(ms, ref, ms, mov, Ts)
    assert np.array_equal(score_r, score) and np.array_equal(nb_r, nb)          # run-to-run: the same bits
    same = [k for k in range(len(ref)) if ref[k] == mov[k] and k % 2 == 0]
    assert np.all(score[same] == 0.0) and np.all(nb[same] > 1000)
    far = np.eye(4); far[0, 3] = 500.0
    s2, n2 = N.overlap_score(ms, [0], ms, [1], far[None])
    assert s2[0] == 1.0 and n2[0] == 0
    # bit-level check of the score arithmetic: the oracle formula on occupancies copied from the GPU
    occ = [ms.occupancy(nd) for nd in range(n_nodes)]
    for k in (1, 7, 23, 40):
        a, b = ref[k], mov[k]
        sx, sy, sz = occ[b].shape
        with np.errstate(over="ignore"):     # exp(255) overflows float32 to inf: 1 - 1/inf = 1, like the C code
            resc = lambda o: np.float32(1) - np.float32(1) / (np.float32(1) + np.exp(o.astype(np.float64)).astype(np.float32))
            om_, or_ = resc(occ[b]), resc(occ[a])
        ii = np.argwhere(om_ != np.float32(0.5))
        ctr = ((ii - np.array([sx // 2, sy // 2, sz // 2])) * 0.5).astype(np.float32).astype(np.float64)
        tp = (ctr @ Ts[k][:3, :3].T + Ts[k][:3, 3]).astype(np.float32).astype(np.float64)
        jj = (np.floor(tp / 0.5 + 0.5) + np.array([sx, sy, sz]) / 2.0).astype(np.int64)
        ok = np.all((jj >= 0) & (jj < np.array([sx, sy, sz])), axis=1)
        ro = or_[jj[ok, 0], jj[ok, 1], jj[ok, 2]]
        mo = om_[ii[ok, 0], ii[ok, 1], ii[ok, 2]]
        use = ro != np.float32(0.5)
        want_nb = int(use.sum())
        want = float(np.sum((mo[use].astype(np.float64) - ro[use].astype(np.float64)) ** 2) / want_nb)
        assert nb[k] == want_nb and abs(score[k] - want) < 1e-12 * max(want, 1e-6), (k, score[k], want)


def test_match_on_fused_maps(N, O):
    """The registration the graph layer actually runs (graph.cpp:273): fused node map against fused node map."""
    from ndt_feature_graph_amd import synth
    per_node = 6
    poses = np.array([[0.25 * k, 0.04 * np.sin(k), 0.015 * k] for k in range(2 * per_node)])
    scans = synth.scan_2d([13] * len(poses), poses, 30000).numpy()
    Tw = synth.pose2d_to_T(poses).numpy()
    ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=2)
    ms.enable_occupancy()
    oms = [O.OracleMap(0.5, [0, 0, 0], [100, 100, 1]) for _ in range(2)]
    for j in range(per_node):
        cl, org = [], []
        for nd in range(2):
            k = nd * per_node + j
            Tl = np.linalg.inv(Tw[nd * per_node]) @ Tw[k]
            cl.append((scans[k].astype(np.float64) @ Tl[:3, :3].T + Tl[:3, 3]).astype(np.float32))
            org.append(Tl[:3, 3])
            oms[nd].add_point_cloud(org[-1], cl[-1], maxz=25.0, sensor_noise=0.06, order_free=True)
            oms[nd].compute_cells_full()
        ms.add_cloud(np.stack(cl), np.stack(org))
    gt = np.linalg.inv(Tw[0]) @ Tw[per_node]
    T0 = gt.copy(); T0[0, 3] += 0.08; T0[1, 3] -= 0.05
    T, r = N.match_d2d(ms, 0, ms, 1, T0, delta_score=1e-3)
    To, ro = O.match_d2d(oms[0], oms[1], T0, delta_score=1e-3)
    assert np.linalg.norm(T[:3, 3] - To[:3, 3]) <= 1e-4 and rot_angle(T[:3, :3], To[:3, :3]) <= 1e-4
    assert r["iterations"] == ro["iterations"] and r["n_target"] == oms[0].num_cells()
    assert np.linalg.norm(T[:3, 3] - gt[:3, 3]) < 0.05


def test_fuse_3d_parity(N, O):
    """3D clouds (Velodyne-style, reduced), 0.5 m voxels, 100 x 100 x 10 m: three sweeps from nearby poses."""
    from ndt_feature_graph_amd import synth
    import torch
    offs = np.array([[0, 0, 0, 0, 0, 0], [0.3, 0.1, 0.02, 0.005, -0.004, 0.03], [0.6, 0.15, 0.03, 0.01, -0.01, 0.06]])
    Ts = synth.pose6_to_T(torch.tensor(offs)).numpy()
    clouds, origins = [], []
    for k in range(3):
        sc = synth.scan_3d([5], torch.tensor(Ts[k:k + 1]), rings=24, azimuths=900, noise_stream=k).numpy()[0]
        clouds.append((sc.astype(np.float64) @ Ts[k][:3, :3].T + Ts[k][:3, 3]).astype(np.float32))
        origins.append(Ts[k][:3, 3])
    ms, om = fuse_both(N, O, clouds, np.array(origins), 0.5, [100, 100, 10], max_cells=60000)
    assert om.num_cells() > 500


def test_discard_cells(N, O):
    """ndt_feature::discardCell (utils.h:229-236): the cells holding the given points lose their Gaussian, everything else
    (cell order, records, the matcher's rank structures) stays consistent."""
    from ndt_feature_graph_amd import synth
    pr = synth.pair_2d([6], 30000)
    f, m = pr["fixed"][0].numpy(), pr["moving"][0].numpy()
    ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=2)
    ms.build(np.stack([f, m]), range_limit=30.0)
    before = ms.export_cells(0)
    ok = ~np.isnan(f[:, 0])
    pts = f[ok][[0, -1, 1234, 1235, 20000]]
    pts = np.concatenate([pts, [[1e3, 0, 0], before[0][7] + 0.01]]).astype(np.float32)     # one outside the grid, one by cell mean
    ms.discard_cells(0, pts)
    after = ms.export_cells(0)
    idx_of = lambda p: tuple((np.floor(p.astype(np.float64) / 0.5 + 0.5) + np.array([100, 100, 1])).astype(int))
    gone = {idx_of(p) for p in pts}
    keep = np.array([tuple(i) not in gone for i in before[2]])
    assert 1 <= (~keep).sum() <= len(pts) and len(after[3]) == keep.sum()
    for x, y in zip(after, before):
        assert np.array_equal(x, y[keep])
    # the matcher sees exactly the remaining cells: derivatives equal the oracle's on a map with the same cells
    om = O.OracleMap(0.5, [0, 0, 0], [100, 100, 1])
    om.set_cells(after[0], after[1])
    mean, cov, _, _ = ms.export_cells(1)
    s1, g1, H1 = N.derivatives(ms, 0, mean, cov)
    so, go, Ho = O.derivatives(om, mean, cov)
    assert abs(s1 - so) < 1e-9 * abs(so) and np.max(np.abs(H1 - Ho)) < 1e-9 * np.max(np.abs(Ho))
    T, r = N.match_d2d(ms, 0, ms, 1, pr["T_init"][0].numpy())
    assert r["n_target"] == keep.sum() and np.all(np.isfinite(T))
