"""BASELINE.json configs at FULL size through the C-ABI, on a real MI355X (-m gpu).

configs[2]  1024 independent pairs x 100 k points, 0.5 m cells, one ndtgpu_match_batch_device call
configs[4]  3D: 200 k-point Velodyne-style clouds, 0.25 m voxels, 100 x 100 x 10 m (6.4 M slots), 6-DoF
configs[3]  replay: >= 200 node maps, all-pairs candidate edges (19 900), block-cyclic shards of 8 ranks
plus the 3-DoF matcher (NDTMatcherD2D_2D) on pairs where it does NOT converge.

The oracle is run on samples (it needs ~20 ms per 2D pair); every pair is covered by properties that need no
oracle: counters consistent, bit-identical re-run, independence of a pair's result from its batch position."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4
POSE_TOL_RAD = 1e-4
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


def rot_angle(Ra, Rb):
    return float(2.0 * np.arcsin(min(1.0, np.linalg.norm(Ra - Rb) / (2.0 * np.sqrt(2.0)))))


def pose_close(Ta, Tb):
    return np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]), rot_angle(Ta[:3, :3], Tb[:3, :3])


def oracle_map(O, pts, res, size, rng, centre=(0, 0, 0)):
    m = O.OracleMap(res, centre, size)
    m.load_points(pts, rng)
    m.compute_cells()
    return m


def cells_equal(gpu, cpu, res):
    gm, gc, gi, gn = gpu
    cm, cc, ci, cn = cpu
    assert len(gn) == len(cn), "number of Gaussian cells differs: %d vs %d" % (len(gn), len(cn))
    assert np.array_equal(gi, ci), "cell index sets differ"
    assert np.array_equal(gn.astype(np.int64), cn.astype(np.int64)), "per-cell point counts differ"
    assert np.max(np.abs(gm - cm)) < 1e-9 * max(1.0, res)
    scale = np.max(np.abs(cc), axis=(1, 2), keepdims=True)
    assert np.max(np.abs(gc - cc) / scale) < 1e-8


def test_config3_batch_1024_pairs_100k_points(N, O):
    """configs[2] through the TWO-CALL path (ndtgpu_mapset_build, then ONE ndtgpu_match_batch_device): scans resident in HBM,
    two 1024-map sets.  bench.py times the one-call registrar instead; that entry has its own full-size test
    (tests/test_gpu_registrar_fullsize.py), which asserts the bits of this path."""
    import torch
    from ndt_feature_graph_amd import binding, synth
    dev = torch.device("cuda", 0)
    B, NP, res, size, rng = 1024, 100000, 0.5, [100.0, 100.0, 1.0], 30.0
    seeds = torch.arange(1, 1 + B, dtype=torch.int64, device=dev)
    pr = synth.pair_2d(seeds, NP, device=dev, chunk_bytes=2 << 30)
    fixed, moving = pr["fixed"].contiguous(), pr["moving"].contiguous()
    T0_cm = pr["T_init"].transpose(1, 2).contiguous().reshape(B, 16)
    tset = N.MapSet(res, [0, 0, 0], size, n_maps=B, max_cells=4096)
    sset = N.MapSet(res, [0, 0, 0], size, n_maps=B, max_cells=4096)
    st = torch.cuda.current_stream()
    tset.build(fixed, range_limit=rng, stream=st)
    sset.build(moving, range_limit=rng, stream=st)

    def run(order):
        idx = torch.as_tensor(order, dtype=torch.int32, device=dev)
        T16 = T0_cm[idx.long()].clone()
        results = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
        binding.match_batch_device(tset, idx, sset, idx, T16, results, B, stream=st)
        torch.cuda.synchronize()
        r = results.cpu().numpy().view(binding.RESULT_DTYPE).reshape(B)
        return T16.cpu().numpy().reshape(B, 4, 4).transpose(0, 2, 1), r

    ident = np.arange(B)
    Ta, ra = run(ident)
    # counters of all 1024 registrations are consistent with the maps and with each other
    nt, ns = tset.num_cells_all(), sset.num_cells_all()
    assert np.array_equal(ra["n_target"], nt) and np.array_equal(ra["n_source"], ns)
    assert nt.min() > 50 and nt.max() < 4096
    assert np.all(ra["iterations"] >= 1) and np.all(ra["iterations"] <= 32)        # ITR_MAX 30 + the reference's overshoot
    assert np.all((ra["exit_code"] == 3) == (ra["converged"] == 0))
    assert np.all(ra["fevals"] >= ra["iterations"]) and np.all(np.isfinite(ra["score"])) and np.all(ra["score"] < 0)
    assert np.all(ra["pair_terms_h"] > 0)
    assert np.all(np.isfinite(Ta)) and np.max(np.abs(np.linalg.det(Ta[:, :3, :3]) - 1)) < 1e-9
    assert ra["converged"].mean() > 0.9
    for k in (0, 511, 1023):
        c = tset.counters(k)
        # points of cells below n_min are binned but in no Gaussian: <=
        assert c["overflow"] == 0 and 0 < tset.export_cells(k)[3].sum() <= NP - c["n_dropped"]
    # run-to-run identical (more pairs than CUs: tickets, parking and resumption in a different order every run)
    Tb, rb = run(ident)
    assert np.array_equal(Ta, Tb)
    for f in DET_FIELDS:
        assert np.array_equal(ra[f], rb[f]), f
    # a pair's result does not depend on where it sits in the batch
    perm = np.random.default_rng(0).permutation(B)
    Tc, rc = run(perm)
    assert np.array_equal(Tc, Ta[perm])
    for f in DET_FIELDS:
        assert np.array_equal(rc[f], ra[f][perm]), f
    # 32 sampled pairs against the CPU matcher (maps bit-equal, poses within the contract's tolerance); the sample
    # takes the longest registrations as well
    sample = sorted(set(list(np.linspace(0, B - 1, 24).astype(int)) + list(np.argsort(-ra["fevals"])[:8])))
    f_h, m_h = fixed[sample].cpu().numpy(), moving[sample].cpu().numpy()
    Ti = pr["T_init"][sample].cpu().numpy()
    for j, b in enumerate(sample):
        ot = oracle_map(O, f_h[j], res, size, rng)
        os_ = oracle_map(O, m_h[j], res, size, rng)
        if j < 4:
            cells_equal(tset.export_cells(b), ot.export_cells(), res)
            cells_equal(sset.export_cells(b), os_.export_cells(), res)
        To, ro = O.match_d2d(ot, os_, Ti[j])
        dt, dr = pose_close(Ta[b], To)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (b, dt, dr)
        assert bool(ra["converged"][b]) == ro["converged"] and ra["iterations"][b] == ro["iterations"], b
        assert abs(ra["score"][b] - ro["score"]) < 1e-6 * abs(ro["score"])


def test_config5_3d_full_size(N, O):
    """configs[4]: 200 k points per cloud, 0.25 m voxels, 400 x 400 x 40 slots (split build + 32-way finalise of a
    6.4 M-slot grid, max_cells 120 000), full 6-DoF registration."""
    from ndt_feature_graph_amd import synth
    pr = synth.pair_3d([1])
    size, res, rng = [100.0, 100.0, 10.0], 0.25, 70.0
    pts = np.concatenate([pr["fixed"].numpy(), pr["moving"].numpy()])
    assert pts.shape[1] == 200000
    ms = N.MapSet(res, [0, 0, 0], size, n_maps=2, max_cells=120000)
    assert ms.info()["cells_per_axis"] == [400, 400, 40]
    ms.build(pts, range_limit=rng)
    of = oracle_map(O, pts[0], res, size, rng)
    om = oracle_map(O, pts[1], res, size, rng)
    assert of.num_cells() > 5000
    cells_equal(ms.export_cells(0), of.export_cells(), res)
    cells_equal(ms.export_cells(1), om.export_cells(), res)
    for k in range(2):
        c = ms.counters(k)
        assert c["overflow"] == 0
    T0 = pr["T_init"][0].numpy()
    T, r = N.match_d2d(ms, 0, ms, 1, T0)
    To, ro = O.match_d2d(of, om, T0)
    dt, dr = pose_close(T, To)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (dt, dr)
    assert bool(r["converged"]) == ro["converged"] and r["iterations"] == ro["iterations"]
    assert pose_close(T, pr["T_gt"][0].numpy())[0] < 0.05
    # a second call on the same set: the cooperative launch (96 workgroups = 16 barrier groups here) must find its
    # control block clean -- a counter left over from the first call sent the registration to the bounded-spin
    # fallback (2.5 s instead of 2 ms): same answer, same shader clocks
    T2, r2 = N.match_d2d(ms, 0, ms, 1, T0)
    assert np.array_equal(T2, T) and r2["fevals"] == r["fevals"]
    assert r2["cycles_eval"] < 3 * r["cycles_eval"], (r["cycles_eval"], r2["cycles_eval"])
    # rebuilding in place gives the same bits (scratch structures return to their clean state)
    before = ms.export_cells(0)
    ms.build(pts, range_limit=rng)
    after = ms.export_cells(0)
    for x, y in zip(before, after):
        assert np.array_equal(x, y)


def _loop_poses(n):
    t = np.linspace(0.0, 2.0 * np.pi, n, endpoint=False)
    return np.stack([1.6 * np.sin(t), 1.2 * np.sin(2.0 * t + 0.3), 0.35 * np.sin(3.0 * t)], axis=1)


def test_config4_replay_200_nodes_sharded(N, O):
    """configs[3] at a size where the machinery matters: 200 node maps, ALL 19 900 candidate edges in
    NDTFeatureGraph::computeAllPossibleLinks order, dealt block-cyclically (chunk 256) to 8 shards, every shard one
    batched call, results reassembled in edge order; 500 sampled edges against the CPU matcher; the reference's
    link gates select a subset that is registered again on its own and must give the same bits."""
    import torch
    from ndt_feature_graph_amd import distributed as D, synth
    dev = torch.device("cuda", 0)
    n_nodes, n_pts, res, size, rng = 200, 30000, 0.5, [100.0, 100.0, 1.0], 30.0
    poses = _loop_poses(n_nodes)
    scans_d = synth.scan_2d(torch.full((n_nodes,), 321, dtype=torch.int64, device=dev),
                            torch.as_tensor(poses, device=dev), n_pts)
    pool = N.MapSet(res, [0, 0, 0], size, n_maps=n_nodes, max_cells=4096)
    pool.build(scans_d.contiguous(), range_limit=rng)
    scans = scans_d.cpu().numpy()
    node_T = synth.pose2d_to_T(poses).numpy()
    g = np.random.default_rng(11)
    odo_T = node_T.copy()
    odo_T[:, 0, 3] += g.normal(scale=0.03, size=n_nodes)
    odo_T[:, 1, 3] += g.normal(scale=0.03, size=n_nodes)
    edges = D.all_pairs(n_nodes)
    assert len(edges) == 19900
    T0 = np.einsum("eij,ejk->eik", np.linalg.inv(odo_T[edges[:, 0]]), odo_T[edges[:, 1]])
    world, chunk = 8, 256
    T_all = np.zeros((len(edges), 4, 4))
    r_all = None
    sizes = D.shard_sizes(len(edges), world, chunk)
    assert sum(sizes) == len(edges) and max(sizes) - min(sizes) <= chunk
    for rank in range(world):
        mine = D.shard_edges(len(edges), rank, world, chunk)
        Tm, rm = N.match_batch(pool, edges[mine, 0], pool, edges[mine, 1], T0[mine], delta_score=1e-3)   # "edge" preset
        if r_all is None:
            r_all = np.zeros(len(edges), dtype=rm.dtype)
        T_all[mine], r_all[mine] = Tm, rm
    assert np.all(np.isfinite(T_all)) and r_all["converged"].mean() > 0.9
    omaps = {}

    def omap(k):
        if k not in omaps:
            omaps[k] = oracle_map(O, scans[k], res, size, rng)
        return omaps[k]
    sample = g.choice(len(edges), 500, replace=False)
    worst = (0.0, 0.0)
    for e in sample:
        i, j = edges[e]
        To, ro = O.match_d2d(omap(i), omap(j), T0[e], delta_score=1e-3)
        dt, dr = pose_close(T_all[e], To)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD and r_all["iterations"][e] == ro["iterations"], (e, dt, dr)
        assert bool(r_all["converged"][e]) == ro["converged"]
        worst = (max(worst[0], dt), max(worst[1], dr))
    # gated subset (NDTFeatureGraph::getValidLinks defaults of ndt_feature_graph_opt.cpp:49-52), registered on its own
    keep = D.gate_links(edges, node_T, max_dist=1.0, max_angle=0.2, min_idx_dist=2)
    assert 100 < len(keep) < len(edges)
    Tg, rg = N.match_batch(pool, edges[keep, 0], pool, edges[keep, 1], T0[keep], delta_score=1e-3)
    assert np.array_equal(Tg, T_all[keep])
    for f in DET_FIELDS:
        assert np.array_equal(rg[f], r_all[f][keep]), f
    # gated edges are between nearby poses: the registration recovers the true relative pose
    gt = np.einsum("eij,ejk->eik", np.linalg.inv(node_T[edges[keep, 0]]), node_T[edges[keep, 1]])
    err = np.array([pose_close(Tg[k], gt[k])[0] for k in range(len(keep))])
    assert np.median(err) < 0.02


def test_3dof_matcher_on_non_converging_pairs(N, O):
    """NDTMatcherD2D_2D ({x, y, yaw}) from the standard synthetic offset: the reference's regulariser
    (lambda_min < 0 => H + (0.001 lambda_max - lambda_min) I on the 3 x 3 block, fusion.h:922-940) makes steps ~1000x
    too long, the registration wanders and finally rolls back to its best pose, which is the initial one (DESIGN.md
    section 7).  The wandering is chaotic: a different summation order (HIP vs oracle, even HIP vs HIP on another
    execution shape) changes the number of iterations before the rollback -- measured, 4 of 24 pairs.  What is pinned:
    the RESULT of every pair (pose within the contract's tolerance, score), bit-identical HIP re-runs, and the control
    flow of every pair on which the ORACLE ITSELF is stable: a control-flow difference is accepted only on a pair where the
    oracle, adding the same pair terms in another order or with its sums moved by a few ulp (oracle_set_sum_mode), itself
    takes another number of iterations or another exit; on every other pair the HIP path must take exactly the oracle's flow, and every control-flow difference must be on a chaotic pair
    where both sides rolled back to the initial pose."""
    from ndt_feature_graph_amd import synth
    seeds = list(range(1, 25))
    pr = synth.pair_2d(seeds, 20000)
    B = len(seeds)
    tg = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=B)
    sr = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=B)
    tg.build(pr["fixed"].numpy(), range_limit=30.0)
    sr.build(pr["moving"].numpy(), range_limit=30.0)
    T0 = pr["T_init"].numpy()
    T, r = N.match_batch(tg, np.arange(B), sr, np.arange(B), T0, dof_mask=0x23)
    T2, r2 = N.match_batch(tg, np.arange(B), sr, np.arange(B), T0, dof_mask=0x23)
    assert np.array_equal(T, T2) and all(np.array_equal(r[f], r2[f]) for f in DET_FIELDS)
    stuck = diverged = chaotic_pairs = 0
    for b in range(B):
        ot = oracle_map(O, pr["fixed"][b].numpy(), 0.5, [100, 100, 1], 30.0)
        os_ = oracle_map(O, pr["moving"][b].numpy(), 0.5, [100, 100, 1], 30.0)
        To, ro = O.match_d2d(ot, os_, T0[b], dof_mask=0x23)
        dt, dr = pose_close(T[b], To)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (b, dt, dr)
        assert abs(r["score"][b] - ro["score"]) < 1e-8 * abs(ro["score"]), b
        assert abs(T[b][2, 3]) < 1e-15 and abs(T[b][2, 2] - 1) < 1e-15             # z, roll, pitch untouched
        same_flow = (bool(r["converged"][b]) == ro["converged"] and r["iterations"][b] == ro["iterations"]
                     and r["exit_code"][b] == ro["exit_code"])
        # a registration that ends at its first iteration has no wandering to be chaotic about: the oracle's flow, always
        assert same_flow or ro["iterations"] > 1, (b, r["iterations"][b], ro["iterations"])
        if not same_flow:
            # "chaotic" as a measurement, not a statement: the ORACLE ALONE must change its control flow on this pair when
            # the SAME pair terms are added in another order (reversed; eight shares i mod 8 added in share order -- the
            # shape of the HIP sums; shares reversed inside), when its 28 sums are moved by a few ulp or its Newton increments by cond(H) eps (what any other
            # arithmetic of the same formulas does; the regularised 3 x 3 system amplifies that by its condition number
            # into the wild steps described above).  A difference on a pair whose oracle flow is stable under all of
            # these would be a discrepancy of the HIP path, not chaos.
            base = (ro["iterations"], ro["exit_code"], ro["converged"])
            chaotic = False
            for mode in list(range(16, 28)) + list(range(1, 16)):
                O.set_sum_mode(mode)
                try:
                    Tq, rq = O.match_d2d(ot, os_, T0[b], dof_mask=0x23)
                finally:
                    O.set_sum_mode(0)
                assert pose_close(Tq, To)[0] <= POSE_TOL_M            # (whatever the flow, the result is the same pose)
                if (rq["iterations"], rq["exit_code"], rq["converged"]) != base:
                    chaotic = True
                    break
            chaotic_pairs += chaotic
            assert chaotic, (b, "the oracle's control flow is stable under other summation orders and ulp noise, the HIP path differs")
        if not same_flow:
            diverged += 1
            assert pose_close(T[b], T0[b])[0] < 1e-12 and pose_close(To, T0[b])[0] < 1e-12, b   # both rolled back
        stuck += pose_close(T[b], pr["T_gt"][b].numpy())[0] > 0.05
    assert stuck >= 1, "expected pairs on which the 3-DoF matcher does not reach the optimum (DESIGN.md section 7)"
    assert diverged == chaotic_pairs and diverged <= B // 3, (diverged, chaotic_pairs)


def test_task_pool_32_pairs_of_full_size_3d_maps(N, O, monkeypatch):
    """configs[4] as a BATCH: 32 pairs of 200 k-point sweeps (12 k cells per map, 0.25 m voxels, 400 x 400 x 40 slots)
    through ndtgpu_match_batch_device -- more than 8 registrations of large maps, i.e. the task pool
    (ndt_match_pool_kernel<2>: any workgroup serves any (registration, evaluation, chunk) task).  Five sampled pairs,
    among them the one with the most evaluations, against the CPU oracle: identical cell sets and point counts, pose,
    iterations; all 32 against the persistent kernel (one CU per registration); a second run gives the same bits."""
    import torch
    from ndt_feature_graph_amd import binding, synth
    dev = torch.device("cuda", 0)
    n = 32
    pr = synth.pair_3d(torch.arange(1, n + 1, device=dev), device=dev)
    sweeps = torch.cat([pr["fixed"], pr["moving"]]).contiguous()
    res, size, rng = 0.25, [100.0, 100.0, 10.0], 70.0
    ms = N.MapSet(res, [0, 0, 0], size, n_maps=2 * n, max_cells=120000)
    ms.build(sweeps, range_limit=rng)
    T0 = pr["T_init"].transpose(1, 2).contiguous().reshape(n, 16)
    ti = torch.arange(n, dtype=torch.int32, device=dev)
    si = ti + n
    runs = {}
    for tag, env in (("pool", {}), ("pool2", {}), ("persistent", {"NDTGPU_DEVICE_COOP": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        T16 = T0.clone()
        out = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
        binding.match_batch_device(ms, ti, ms, si, T16, out, n, stream=torch.cuda.current_stream())
        torch.cuda.synchronize()
        runs[tag] = (T16.cpu().numpy().reshape(n, 4, 4).transpose(0, 2, 1).copy(),
                     out.cpu().numpy().view(binding.RESULT_DTYPE).reshape(n).copy())
        for k in env:
            monkeypatch.delenv(k)
    T, r = runs["pool"]
    assert np.all(r["exit_code"] >= 0) and r["converged"].mean() > 0.9
    assert np.array_equal(T, runs["pool2"][0]) and all(np.array_equal(r[f], runs["pool2"][1][f]) for f in DET_FIELDS)
    Tp, rp = runs["persistent"]
    assert np.max(np.abs(T - Tp)) < 1e-7 and np.array_equal(r["iterations"], rp["iterations"])      # another summation order
    longest = int(np.argmax(r["fevals"]))
    sample = sorted({longest, 0, 7, 19, int(np.argmin(r["fevals"]))})
    assert len(sample) >= 4
    cells = ms.num_cells_all()
    assert cells.min() > 8000
    for k in sample:
        fx, mv = sweeps[k].cpu().numpy(), sweeps[n + k].cpu().numpy()
        a, b = oracle_map(O, fx, res, size, rng), oracle_map(O, mv, res, size, rng)
        for m, om in ((k, a), (n + k, b)):
            g, o = ms.export_cells(m), om.export_cells()
            assert np.array_equal(g[2], o[2]) and np.array_equal(g[3], o[3]), (k, m)              # cell sets, point counts: exact
        To, ro = O.match_d2d(a, b, pr["T_init"][k].cpu().numpy())
        dt, dr = pose_close(T[k], To)
        assert dt <= 1e-6 and dr <= 1e-6, (k, dt, dr)
        assert r["iterations"][k] == ro["iterations"] and bool(r["converged"][k]) == ro["converged"], k
