"""Randomised parity campaigns as part of the GPU suite (slices of tools/parity_campaign.py / parity_campaign_3d.py):
many seeds, HIP path (grid build + D2D matcher) against the C oracle.  Tolerance of the metric: 1e-4 m / 1e-4 rad
(BASELINE.json north_star); what is asserted here is far tighter and includes the control flow (iteration counts,
convergence flags)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
POSE_TOL_M, POSE_TOL_RAD = 1e-4, 1e-4


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


def test_parity_campaign_2d_300_pairs(N, O):
    """300 random 2D pairs (seeds 5000-5299, 20 k points, 0.5 m cells) through the persistent matcher."""
    import torch
    from ndt_feature_graph_amd import synth
    n, npts, res, size, rng = 300, 20000, 0.5, [100.0, 100.0, 1.0], 30.0
    seeds = list(range(5000, 5000 + n))
    pr = synth.pair_2d(seeds, npts)
    fixed, moving, T0 = pr["fixed"].numpy(), pr["moving"].numpy(), pr["T_init"].numpy()
    ms = N.MapSet(res, [0, 0, 0], size, n_maps=2 * n, max_cells=4096)
    ms.build(torch.from_numpy(np.concatenate([fixed, moving])).cuda(), range_limit=rng)
    torch.cuda.synchronize()
    idx = np.arange(n)
    T, r = N.match_batch(ms, idx, ms, idx + n, T0)
    worst_t = worst_r = 0.0
    it_diff = conv_diff = beyond = 0
    for k in range(n):
        a = O.OracleMap(res, [0, 0, 0], size); a.load_points(fixed[k], rng); a.compute_cells()
        b = O.OracleMap(res, [0, 0, 0], size); b.load_points(moving[k], rng); b.compute_cells()
        assert r["n_target"][k] == a.num_cells() and r["n_source"][k] == b.num_cells(), seeds[k]
        To, ro = O.match_d2d(a, b, T0[k])
        dt = float(np.linalg.norm(T[k][:3, 3] - To[:3, 3]))
        dr = float(np.linalg.norm(T[k][:3, :3] - To[:3, :3]))
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (seeds[k], dt, dr)
        worst_t, worst_r = max(worst_t, dt), max(worst_r, dr)
        it_diff += int(r["iterations"][k] != ro["iterations"])
        conv_diff += int(bool(r["converged"][k]) != ro["converged"])
        beyond += int(dt > 1e-6 or dr > 1e-6)
    print("2D campaign: worst |dt| %.3e m |dR| %.3e, iteration counts differ on %d, beyond 1e-6 on %d" % (worst_t, worst_r, it_diff, beyond))
    assert conv_diff == 0
    # a step norm that sits on DELTA_SCORE can cost a pair an iteration more or less (1 of 1500 at the end of round 2)
    assert it_diff <= 2 and beyond <= 2, (it_diff, beyond)


def test_parity_campaign_3d_8_pairs(N, O):
    """8 random pairs of 48 k-point 3D sweeps at 0.25 m (thick grid: wide flush list, multi-workgroup ranking,
    cooperative matcher): identical cell sets and point counts on all 16 maps, poses, iteration counts."""
    from ndt_feature_graph_amd import synth
    n = 8
    seeds = list(range(300, 300 + n))
    pr = synth.pair_3d(seeds, rings=32, azimuths=1500)
    fixed, moving, T0 = pr["fixed"].numpy(), pr["moving"].numpy(), pr["T_init"].numpy()
    res, size, rng = 0.25, [100.0, 100.0, 10.0], 70.0
    ms = N.MapSet(res, [0, 0, 0], size, n_maps=2 * n, max_cells=32768)
    ms.build(np.concatenate([fixed, moving]), range_limit=rng)
    idx = np.arange(n)
    T, r = N.match_batch(ms, idx, ms, idx + n, T0)
    for k in range(n):
        a = O.OracleMap(res, [0, 0, 0], size); a.load_points(fixed[k], rng); a.compute_cells()
        b = O.OracleMap(res, [0, 0, 0], size); b.load_points(moving[k], rng); b.compute_cells()
        for m, om in ((k, a), (n + k, b)):
            g, o = ms.export_cells(m), om.export_cells()
            assert np.array_equal(g[2], o[2]) and np.array_equal(g[3], o[3]), (seeds[k], m)
        To, ro = O.match_d2d(a, b, T0[k])
        dt = float(np.linalg.norm(T[k][:3, 3] - To[:3, 3]))
        dr = float(np.linalg.norm(T[k][:3, :3] - To[:3, :3]))
        assert dt <= 1e-6 and dr <= 1e-6, (seeds[k], dt, dr)
        assert r["iterations"][k] == ro["iterations"] and bool(r["converged"][k]) == ro["converged"], seeds[k]


def test_3d_pair_that_runs_into_itr_max(N, O):
    """Seed 371 of the 3D campaign does not converge: 32 iterations and exit "too many iterations" on both sides, and the
    two poses are 3.7e-4 m apart.  What is asserted: the control flow is the oracle's (iterations, exit, convergence
    flag), the maps are identical, and the HIP path's pose is the ORACLE's pose under one of its summation orders
    (oracle_set_sum_mode 0..3: default, reversed, eight shares, shares reversed) to 1e-6 m -- a registration that does
    not converge has no pose to agree on beyond the order in which its pair terms are added."""
    from ndt_feature_graph_amd import synth
    pr = synth.pair_3d([371], rings=32, azimuths=1500)
    fixed, moving, T0 = pr["fixed"].numpy(), pr["moving"].numpy(), pr["T_init"].numpy()
    res, size, rng = 0.25, [100.0, 100.0, 10.0], 70.0
    ms = N.MapSet(res, [0, 0, 0], size, n_maps=2, max_cells=32768)
    ms.build(np.concatenate([fixed, moving]), range_limit=rng)
    T, r = N.match_batch(ms, [0], ms, [1], T0)
    a = O.OracleMap(res, [0, 0, 0], size); a.load_points(fixed[0], rng); a.compute_cells()
    b = O.OracleMap(res, [0, 0, 0], size); b.load_points(moving[0], rng); b.compute_cells()
    for m, om in ((0, a), (1, b)):
        g, o = ms.export_cells(m), om.export_cells()
        assert np.array_equal(g[2], o[2]) and np.array_equal(g[3], o[3])
    dts = []
    try:
        for mode in (0, 1, 2, 3):
            O.set_sum_mode(mode)
            To, ro = O.match_d2d(a, b, T0[0])
            assert r["iterations"][0] == ro["iterations"] and bool(r["converged"][0]) == bool(ro["converged"])
            dts.append(float(np.linalg.norm(T[0][:3, 3] - To[:3, 3])))
    finally:
        O.set_sum_mode(0)
    print("3D pair 371: |dt| to the oracle under summation orders 0..3:", ["%.2e" % d for d in dts])
    assert not r["converged"][0] and r["exit_code"][0] == 3
    assert min(dts) <= 1e-6 and max(dts) <= 2e-3, dts


def test_parity_campaign_match_fusion_slice(N, O):
    """A slice of tools/parity_campaign_fusion.py: 40 random 2D pairs (seeds 7160-7199) x the eight combinations of
    {soft constraint, Tikhonov, joint line search}, each with odometry cell pairs and a random odometry covariance, through
    ndtgpu_match_fusion_feat_batch against oracle_match_fusion_feat.  A pair whose pose differs by more than 1e-6 m must be one
    that does not converge (ITR_MAX on both sides) AND whose HIP pose lies within the scatter of the oracle's OWN poses under
    another summation order / ulp noise on its sums / cond(H) eps on its Newton increments (oracle_set_sum_mode 1..27):
    seeds 7173 and 7178 are such pairs (the full campaign: 12 of 4800 registrations)."""
    from ndt_feature_graph_amd import synth, binding
    n, npts, res, size, rng_lim = 40, 20000, 0.5, [100.0, 100.0, 1.0], 30.0
    seeds = list(range(7160, 7160 + n))
    pr = synth.pair_2d(seeds, npts)
    fixed, moving, T0, Tgt = pr["fixed"].numpy(), pr["moving"].numpy(), pr["T_init"].numpy(), pr["T_gt"].numpy()
    tg = N.MapSet(res, [0, 0, 0], size, n_maps=n)
    sr = N.MapSet(res, [0, 0, 0], size, n_maps=n)
    tg.build(fixed, range_limit=rng_lim)
    sr.build(moving, range_limit=rng_lim)
    g = np.random.default_rng(11)
    odom_cov6 = np.array([4e-4, 1e-5, 0.0, 6e-4, 0.0, 0.01])
    covs, feats, om = [], [], []
    for i in range(160 + n):                                  # (the campaign's random stream: this slice is its pairs 160..199)
        A = g.normal(size=(6, 6)) * 0.01
        cov = A @ A.T + np.diag([2e-3, 2e-3, 1.0, 1.0, 1.0, 4e-4]) * (1.0 + g.uniform(0, 1))
        noise = g.normal(scale=0.02, size=2)
        k = 40 if i % 7 else int(g.integers(0, 12))
        if i < 160:
            continue
        b = i - 160
        covs.append(cov)
        Todo = Tgt[b].copy()
        Todo[:2, 3] += noise
        feats.append((np.zeros((k, 3)), np.tile(odom_cov6, (k, 1)), np.tile(Todo[:3, 3], (k, 1)), np.tile(odom_cov6, (k, 1))))
        a = O.OracleMap(res, [0, 0, 0], size); a.load_points(fixed[b], rng_lim); a.compute_cells()
        c = O.OracleMap(res, [0, 0, 0], size); c.load_points(moving[b], rng_lim); c.compute_cells()
        om.append((a, c))
    covs = np.stack(covs)
    idx = np.arange(n)
    loose = 0
    try:
        for soft in (False, True):
            for tik in (False, True):
                for joint in (False, True):
                    kw = dict(use_soft_constraints=soft, tikhonov=tik, step_control_fusion=joint)
                    Tb, rb = binding.match_fusion_feat_batch(tg, idx, sr, idx, T0, covs, feats, **kw)
                    for b in range(n):
                        To, ro = O.binding.match_fusion_feat(om[b][0], om[b][1], T0[b], covs[b], feats[b], **kw)
                        dt = float(np.linalg.norm(Tb[b][:3, 3] - To[:3, 3]))
                        if dt <= 1e-6:
                            assert rb["iterations"][b] == ro["iterations"] and rb["exit_code"][b] == ro["exit_code"], (kw, seeds[b])
                            continue
                        loose += 1
                        assert not rb["converged"][b] and not ro["converged"], (kw, seeds[b], dt)
                        best, spread = dt, 0.0
                        for mode in range(1, 28):
                            O.set_sum_mode(mode)
                            Tm, _ = O.binding.match_fusion_feat(om[b][0], om[b][1], T0[b], covs[b], feats[b], **kw)
                            best = min(best, float(np.linalg.norm(Tb[b][:3, 3] - Tm[:3, 3])))
                            spread = max(spread, float(np.linalg.norm(To[:3, 3] - Tm[:3, 3])))
                        O.set_sum_mode(0)
                        assert best <= max(1e-6, spread), (kw, seeds[b], dt, best, spread)
    finally:
        O.set_sum_mode(0)
    print("matchFusion slice: %d of %d registrations beyond 1e-6 m, all non-converging and within the oracle's own scatter" % (loose, 8 * n))
    assert 1 <= loose <= 8
