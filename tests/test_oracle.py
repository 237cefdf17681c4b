"""CPU oracle (oracle/ndt_oracle.c) pinned against the committed NumPy golden vectors
(tests/golden/make_golden.py) and against first-principles properties.  The reference holds no
vectors of its own for this path -- these are the authored KATs K1-K7."""
import numpy as np
import pytest

import oracle as O


def _mk(res, centre, size_cells):
    return O.OracleMap(res, centre, np.asarray(size_cells, float) * res)


# ---- K6 -------------------------------------------------------------------------------------
@pytest.mark.parametrize("k", range(4))
def test_index_for_point(golden, k):
    res = float(golden["idx%d_res" % k])
    m = _mk(res, golden["idx%d_centre" % k], golden["idx%d_size" % k])
    size = golden["idx%d_size" % k]
    for p, want in zip(golden["idx%d_pts" % k], golden["idx%d_idx" % k]):
        got, inside = m.index_for_point(p)
        assert got == list(want)
        assert inside == bool(np.all(want >= 0) and np.all(want < size))


# ---- K5 -------------------------------------------------------------------------------------
@pytest.mark.parametrize("k", range(6))
def test_cell_gaussian(golden, k):
    pts = golden["cell%d_pts" % k]
    centre = np.round(pts.mean(axis=0).astype(float) * 2) / 2
    m = O.OracleMap(0.5, centre, [0.5, 0.5, 0.5])       # a single voxel around the cloud
    m.load_points(pts)
    m.compute_cells(n_min=3, eval_factor=1000.0)
    if not bool(golden["cell%d_ok" % k]):
        assert m.num_cells() == 0          # rank-deficient sample covariance: no Gaussian
        return
    assert m.num_cells() == 1
    mean, cov, idx, n = m.export_cells()
    assert n[0] == len(pts)
    np.testing.assert_allclose(mean[0], golden["cell%d_mean" % k], rtol=0, atol=1e-12)
    np.testing.assert_allclose(cov[0], golden["cell%d_cov" % k], rtol=1e-9, atol=1e-15)


def test_cell_min_points_and_degenerate():
    m = O.OracleMap(1.0, [0, 0, 0], [4, 4, 1])
    pts = np.array([[0.1, 0.1, 0.0], [0.2, -0.1, 0.01],              # 2 points: below n_min=3
                    [1.1, 1.0, 0.0], [1.2, 1.1, 0.0], [0.9, 0.8, 0.0], [1.0, 1.2, 0.0]], np.float32)  # planar z
    m.load_points(pts)
    m.compute_cells()
    assert m.num_cells() == 0          # first cell too few points; second has a zero eigenvalue
    m.load_points(pts)
    m.compute_cells(n_min=2)
    assert m.num_cells() == 1 or m.num_cells() == 0


def test_load_filters_nan_range_and_outside():
    m = O.OracleMap(1.0, [0, 0, 0], [10, 10, 2])
    base = np.array([[0.1, 0.1, 0.1], [0.2, 0.15, 0.05], [0.15, 0.3, 0.2], [0.3, 0.2, 0.15]], np.float32)
    junk = np.array([[np.nan, 0, 0], [0, np.nan, 0], [4.0, 0.2, 0.1], [100.0, 0, 0], [0, -5.6, 0]], np.float32)
    m.load_points(np.concatenate([base, junk]), range_limit=3.0)
    m.compute_cells()
    mean, cov, idx, n = m.export_cells()
    assert len(n) == 1 and n[0] == 4
    # range measured from an origin (loadPointCloudCentroid semantics)
    m.load_points(np.concatenate([base, junk]), range_limit=3.0, range_origin=[3.0, 0, 0])
    m.compute_cells(n_min=1)
    mean, cov, idx, n = m.export_cells()
    assert sorted(n.tolist()) == [1, 4] or sorted(n.tolist()) == [4]   # (4.0,0.2,0.1) now in range


# ---- K2 -------------------------------------------------------------------------------------
def _d2d_setup(golden):
    res = float(golden["d2d_res"])
    tgt = _mk(res, golden["d2d_centre"], golden["d2d_size"])
    tgt.set_cells(golden["d2d_tgt_mean"], golden["d2d_tgt_cov"])
    return tgt


def test_d2d_score_gradient_hessian_vs_numpy_fd(golden):
    tgt = _d2d_setup(golden)
    s, g, H = O.derivatives(tgt, golden["d2d_src_mean"], golden["d2d_src_cov"], n_neighbours=2)
    assert abs(s - float(golden["d2d_score"])) < 1e-11 * abs(s)
    gfd, Hfd = golden["d2d_grad_fd"], golden["d2d_hess_fd"]
    assert np.max(np.abs(g - gfd)) < 1e-6 * np.max(np.abs(gfd))       # analytic vs central FD
    assert np.max(np.abs(H - Hfd)) < 2e-5 * np.max(np.abs(Hfd))
    np.testing.assert_allclose(H, H.T, rtol=1e-12, atol=1e-12)
    s2, g2, _ = O.derivatives(tgt, golden["d2d_src_mean"], golden["d2d_src_cov"], compute_hessian=False)
    assert s2 == s and np.array_equal(g, g2)


def test_d2d_neighbourhood_sizes(golden):
    tgt = _d2d_setup(golden)
    s0, _, _ = O.derivatives(tgt, golden["d2d_src_mean"], golden["d2d_src_cov"], n_neighbours=0)
    s1, _, _ = O.derivatives(tgt, golden["d2d_src_mean"], golden["d2d_src_cov"], n_neighbours=1)
    s2, _, _ = O.derivatives(tgt, golden["d2d_src_mean"], golden["d2d_src_cov"], n_neighbours=2)
    assert s2 < s1 < s0 <= 0.0          # every extra pair adds a negative term


def test_gradient_fd_on_scan_maps():
    """FD check through the whole stack (maps from points, pseudo-transform, neighbourhood)."""
    import torch
    from ndt_feature_graph_amd import synth
    pr = synth.pair_2d([3], 4000)
    tg = O.OracleMap(1.0, [0, 0, 0], [100, 100, 1]); tg.load_points(pr["fixed"][0].numpy(), 30.0); tg.compute_cells()
    sr = O.OracleMap(1.0, [0, 0, 0], [100, 100, 1]); sr.load_points(pr["moving"][0].numpy(), 30.0); sr.compute_cells()
    T = pr["T_gt"][0].numpy()
    mean, cov, _, _ = sr.export_cells()
    s, g, H = O.derivatives(tg, mean @ T[:3, :3].T + T[:3, 3], T[:3, :3] @ cov @ T[:3, :3].T)
    assert abs(s - O.score_at(tg, sr, T, np.zeros(6))) < 1e-9 * abs(s)
    eps = 1e-6
    for a in range(6):
        e = np.zeros(6); e[a] = eps
        fd = (O.score_at(tg, sr, T, e) - O.score_at(tg, sr, T, -e)) / (2 * eps)
        assert abs(fd - g[a]) < 1e-5 * max(1.0, np.max(np.abs(g)))


# ---- K1 -------------------------------------------------------------------------------------
def test_mahalanobis_newton_step(golden):
    Cm, x, x0 = golden["maha_C"], golden["maha_x"], golden["maha_x0"]
    s, g, H = O.mahalanobis(x - x0, Cm)
    assert abs(s - float(golden["maha_score"])) < 1e-12 * abs(s)
    np.testing.assert_allclose(g, golden["maha_grad"], rtol=1e-13)
    np.testing.assert_allclose(H, golden["maha_hess"], rtol=1e-13)
    dx = -O.ldlt_solve(H, g)
    np.testing.assert_allclose(x + dx, x0, atol=1e-12)            # one Newton step lands on x0
    np.testing.assert_allclose(H @ dx, -g, rtol=1e-12)            # "this should be equal to the gradient"


# ---- K7 -------------------------------------------------------------------------------------
def test_mt_cstep_vectors(golden):
    for row in golden["mt_cstep"]:
        stx, fx, dx, sty, fy, dy, stp, fp, dp, br, stmin, stmax, info = row[:13]
        want = row[13:20]
        got_info, got, got_br = O.mt_cstep(stx, fx, dx, sty, fy, dy, stp, fp, dp, bool(br), stmin, stmax)
        assert got_info == int(info)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
        assert got_br == bool(row[20])


def test_mt_cstep_invalid_input_returns_zero():
    # dx*(stp-stx) >= 0 and stmax < stmin are rejected (info 0), state untouched
    info, st, br = O.mt_cstep(0.0, 1.0, +0.5, 2.0, 1.0, 0.1, 1.0, 0.9, -0.1, False, 0.0, 4.0)
    assert info == 0 and st[6] == 1.0
    info, st, br = O.mt_cstep(0.0, 1.0, -0.5, 2.0, 1.0, 0.1, 1.0, 0.9, -0.1, False, 4.0, 0.0)
    assert info == 0


def test_mt_linesearch_1d(golden):
    import importlib.util, os
    spec = importlib.util.spec_from_file_location(
        "make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    for name, (f, df) in mg.LS_FUNCS.items():
        stp, nfev, info = O.mt_linesearch(lambda t: (float(f(t)), float(df(t))), float(f(0.0)), float(df(0.0)))
        want = golden["mt_ls_" + name]
        assert (nfev, info) == (int(want[1]), int(want[2])), name
        assert abs(stp - want[0]) < 1e-12 * max(1.0, abs(want[0])), name
    # a full quadratic Newton step is accepted at stp = 1 in one evaluation
    assert tuple(golden["mt_ls_quad_unit"]) == (1.0, 1.0, 1.0)


# ---- algebra ----------------------------------------------------------------------------------
def test_small_algebra():
    rng = np.random.default_rng(1)
    for n in (3, 6):
        A = rng.normal(size=(n, n)); A = A @ A.T - 0.5 * np.eye(n)
        ev, V = O.eig_sym(A)
        np.testing.assert_allclose(ev, np.linalg.eigvalsh(A), atol=1e-12)
        np.testing.assert_allclose(V @ np.diag(ev) @ V.T, A, atol=1e-12)
        b = rng.normal(size=n)
        np.testing.assert_allclose(O.ldlt_solve(A, b), np.linalg.solve(A, b), rtol=1e-9)
    p = np.array([0.3, -0.2, 0.1, 0.05, -0.07, 0.4])
    T = O.pose_to_T(p)
    cx, sx, cy, sy, cz, sz = np.cos(p[3]), np.sin(p[3]), np.cos(p[4]), np.sin(p[4]), np.cos(p[5]), np.sin(p[5])
    R = (np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
         @ np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]))
    np.testing.assert_allclose(T[:3, :3], R, atol=1e-15)
    np.testing.assert_allclose(T[:3, 3], p[:3])


# ---- K3 / K4 ------------------------------------------------------------------------------------
def _scan_maps(seed, n, res):
    from ndt_feature_graph_amd import synth
    pr = synth.pair_2d([seed], n)
    maps = []
    for key in ("fixed", "moving"):
        m = O.OracleMap(res, [0, 0, 0], [100, 100, 1])
        m.load_points(pr[key][0].numpy(), 30.0)
        m.compute_cells()
        maps.append(m)
    return maps[0], maps[1], pr["T_init"][0].numpy(), pr["T_gt"][0].numpy()


def test_self_match_is_identity():
    tg, _, _, _ = _scan_maps(5, 5000, 1.0)
    T, r = O.match_d2d(tg, tg, np.eye(4))
    assert r["converged"]
    assert np.max(np.abs(T - np.eye(4))) < 1e-9
    mean, cov, _, _ = tg.export_cells()
    s, g, _ = O.derivatives(tg, mean, cov)
    assert abs(r["score"] - s) < 1e-9 * abs(s)
    assert np.linalg.norm(g) < 1e-6 * abs(s)        # stationary at identity


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_known_transform_recovery_config1(seed):
    """Config 1 (BASELINE.json configs[0]): 10 k points, 1.0 m cells, CPU matcher only."""
    tg, sr, T0, Tgt = _scan_maps(seed, 10000, 1.0)
    T, r = O.match_d2d(tg, sr, T0)
    assert r["converged"] and r["iterations"] <= 31
    dt = np.linalg.norm(T[:3, 3] - Tgt[:3, 3])
    dyaw = abs(np.arctan2(T[1, 0], T[0, 0]) - np.arctan2(Tgt[1, 0], Tgt[0, 0]))
    dt0 = np.linalg.norm(T0[:3, 3] - Tgt[:3, 3])
    assert dt < 0.02 and dyaw < 2e-3          # grid-limited accuracy, far better than the guess
    assert dt < 0.25 * dt0
    # numerical optimality of the returned pose: FD gradient ~ 0 relative to the gradient at T0
    g0 = np.array([(O.score_at(tg, sr, T0, e) - O.score_at(tg, sr, T0, -e)) / 2e-6 for e in 1e-6 * np.eye(6)])
    g1 = np.array([(O.score_at(tg, sr, T, e) - O.score_at(tg, sr, T, -e)) / 2e-6 for e in 1e-6 * np.eye(6)])
    assert np.linalg.norm(g1) < 1e-3 * np.linalg.norm(g0)
    assert O.score_at(tg, sr, T, np.zeros(6)) < O.score_at(tg, sr, T0, np.zeros(6))


def test_iteration_cap_reports_not_converged():
    tg, sr, T0, _ = _scan_maps(1, 10000, 1.0)
    T, r = O.match_d2d(tg, sr, T0, itr_max=1)
    assert (not r["converged"]) and r["iterations"] == 3 and r["exit_code"] == 3   # itr_ctr 0,1,2 then cap


def test_no_step_control_and_no_initial_guess():
    tg, sr, T0, Tgt = _scan_maps(2, 10000, 1.0)
    T, r = O.match_d2d(tg, sr, Tgt, step_control=0)
    assert r["fevals"] == r["iterations"] + 1 or r["exit_code"] in (1, 2)
    Tn, rn = O.match_d2d(tg, sr, T0, use_initial_guess=0)
    Ti, ri = O.match_d2d(tg, sr, np.eye(4))
    np.testing.assert_allclose(Tn, Ti, atol=0)


# ---- matchFusion soft constraint (next row 8f-3) --------------------------------------------------
def test_match_fusion_prior_limits():
    """A weak odometry prior (huge Tcov) reproduces NDTMatcherD2D::match; a strong one (tiny Tcov) keeps
    the pose at the initial guess; without soft constraints matchFusion IS match."""
    tg, sr, T0, Tgt = _scan_maps(3, 10000, 1.0)
    Tm, rm = O.match_d2d(tg, sr, T0)
    Tn, rn = O.match_fusion(tg, sr, T0, np.eye(6), use_soft_constraints=False)
    assert np.array_equal(Tm, Tn) and rm == rn
    Tw, rw = O.match_fusion(tg, sr, T0, 1e12 * np.eye(6))
    assert np.max(np.abs(Tw - Tm)) < 1e-6
    Ts, rs = O.match_fusion(tg, sr, T0, 1e-10 * np.eye(6))
    assert np.max(np.abs(Ts - T0)) < 1e-6
    # in between the prior pulls the solution towards the guess
    Tmid, _ = O.match_fusion(tg, sr, T0, np.diag([1e-4, 1e-4, 1.0, 1.0, 1.0, 1e-5]))
    d_free, d_mid = np.linalg.norm(Tm[:3, 3] - T0[:3, 3]), np.linalg.norm(Tmid[:3, 3] - T0[:3, 3])
    assert 0 < d_mid < d_free


def test_order_free_ray_tracing_equals_the_sequential_walk_on_node_maps():
    """The HIP path applies the beam evidence of a cloud order-free (every beam sees the cells as they were when the call
    started, exact integer sums); the reference walks beam after beam.  The oracle implements both: on 20 node maps of
    16 scans x 20 k points they yield exactly the same Gaussian cells with the same N (what the matcher sees), and
    occupancies within float accumulation error."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_fuse import node_scans
    for seed in range(1, 21):
        clouds, origins, _ = node_scans(seed, 16, 20000)
        maps = []
        for order_free in (True, False):
            om = O.OracleMap(0.5, [0, 0, 0], [100, 100, 1])
            for k in range(len(clouds)):
                kw = dict(maxz=100.0, sensor_noise=0.1) if k == 0 else dict(maxz=25.0, sensor_noise=0.06)
                om.add_point_cloud(origins[k], clouds[k], order_free=order_free, **kw)
                om.compute_cells_full()
            maps.append(om)
        a, b = maps[0].export_cells(), maps[1].export_cells()
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]), seed          # cell indices, N per cell
        assert np.max(np.abs(a[0] - b[0])) < 1e-12 and np.max(np.abs(a[1] - b[1])) < 1e-12, seed
        assert np.max(np.abs(maps[0].occupancy() - maps[1].occupancy())) < 1e-2, seed


def test_summation_order_knob_adds_the_same_terms():
    """oracle_set_sum_mode (the knob tests/test_gpu_fullsize.py measures chaos with): every mode adds the same pair terms
    -- score, gradient and Hessian agree to rounding, and differ in their last bits for at least one mode."""
    from ndt_feature_graph_amd import synth
    pr = synth.pair_2d([3], 20000)
    a = O.OracleMap(0.5, [0, 0, 0], [100, 100, 1]); a.load_points(pr["fixed"][0].numpy(), 30.0); a.compute_cells()
    b = O.OracleMap(0.5, [0, 0, 0], [100, 100, 1]); b.load_points(pr["moving"][0].numpy(), 30.0); b.compute_cells()
    T = pr["T_init"][0].numpy()
    mean, cov, _, _ = b.export_cells()
    m = mean @ T[:3, :3].T + T[:3, 3]
    C = T[:3, :3] @ cov @ T[:3, :3].T
    ref = O.derivatives(a, m, C)
    differs = False
    try:
        for mode in (1, 2, 3):
            O.set_sum_mode(mode)
            got = O.derivatives(a, m, C)
            for x, y in zip(got, ref):
                x, y = np.asarray(x), np.asarray(y)
                assert np.max(np.abs(x - y)) <= 1e-12 * max(1.0, np.max(np.abs(y)))
                differs = differs or not np.array_equal(x, y)
    finally:
        O.set_sum_mode(0)
    assert differs, "another summation order should move the last bits of some sum"
    again = O.derivatives(a, m, C)
    assert all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(again, ref))
