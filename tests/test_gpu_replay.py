"""configs[3] at the survey's CI size on FUSED node maps (-m gpu): 500 nodes 2 m apart through five rooms, every node a
map fused from several ray-traced scans (what ndt_feature_graph.cpp:273 really registers), built data-parallel the way
bench.py --config 4 does it on several ranks -- node k in the set of "rank" k mod world --, exchanged as packed cell
records (ndtgpu_mapset_pack_cells_device -> the reordering of distributed.exchange_node_maps -> _unpack_cells_device) and
registered: the gated candidate edges (ndt_feature_graph_opt.cpp:49-52, :131-160) and ALL pairs of
computeAllPossibleLinks (ndt_feature_graph.cpp:395-405).  300 sampled edges against the CPU oracle: pose, iterations,
covariance (graph.cpp:296-298), occupancy overlap (graph.cpp:338-340; ndt_feature_node.h:213-252)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4       # north_star: output pose per pair within 1e-4 m / 1e-4 rad of the CPU matcher
POSE_TOL_RAD = 1e-4
DET_FIELDS = ("converged", "iterations", "fevals", "exit_code", "score", "n_source", "n_target", "pair_terms_g", "pair_terms_h")


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


def pose_close(Ta, Tb):
    dt = float(np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]))
    dr = float(2.0 * np.arcsin(min(1.0, np.linalg.norm(Ta[:3, :3] - Tb[:3, :3]) / (2.0 * np.sqrt(2.0)))))
    return dt, dr


def replay_layout(n_nodes, per_room=100):
    """The trajectory of bench.py --config 4: 100 nodes per room on a serpentine path, 2 m between consecutive nodes."""
    q = np.arange(n_nodes) % per_room
    room = np.arange(n_nodes) // per_room
    col, row = q // 10, q % 10
    row = np.where(col % 2 == 1, 9 - row, row)
    lx, ly = -9.0 + 2.0 * col, -9.0 + 2.0 * row
    yaw = np.where(col % 2 == 1, -np.pi / 2, np.pi / 2)
    world_xy = np.stack([80.0 * (room % 8) + lx, 80.0 * (room // 8) + ly], axis=1)
    return room, np.stack([lx, ly, yaw], axis=1), np.concatenate([world_xy, yaw[:, None]], axis=1)


def test_replay_500_fused_nodes_sharded_build_and_exchange(N, O):
    import torch
    from ndt_feature_graph_amd import binding, distributed as D, synth
    dev = torch.device("cuda", 0)
    n_nodes, S, n_pts, res, size = 500, 3, 6000, 0.5, [100.0, 100.0, 1.0]
    room, local, world_pose = replay_layout(n_nodes)
    node_T = synth.pose2d_to_T(torch.as_tensor(world_pose)).numpy()
    g = np.random.default_rng(11)
    odo_T = node_T.copy()
    odo_T[:, 0, 3] += g.normal(scale=0.03, size=n_nodes)
    odo_T[:, 1, 3] += g.normal(scale=0.03, size=n_nodes)
    seeds = torch.as_tensor(4000 + room, dtype=torch.int64, device=dev)
    clouds = []
    for k in range(S):
        dx = 0.9 * k / S
        pk = local.copy()
        pk[:, 0] += dx * np.cos(local[:, 2]); pk[:, 1] += dx * np.sin(local[:, 2])
        sc = synth.scan_2d(seeds, torch.as_tensor(pk, device=dev), n_pts, noise_stream=k).contiguous()
        sc[:, :, 0] += dx                                              # sensor frame -> node frame
        clouds.append((sc, np.tile(np.array([[dx, 0.0, 0.0]]), (n_nodes, 1))))
    fuse_kw = [dict(maxz=100.0, sensor_noise=0.1)] + [dict(maxz=25.0, sensor_noise=0.06)] * (S - 1)

    # the reference for the exchange: every node map built in ONE set
    direct = N.MapSet(res, [0, 0, 0], size, n_maps=n_nodes, max_cells=2048)
    direct.enable_occupancy()
    for (sc, org), kw in zip(clouds, fuse_kw):
        direct.add_cloud(sc, org, **kw)
    cells_per_map = direct.num_cells_all()
    assert cells_per_map.min() > 5 and cells_per_map.mean() > 100

    # phases A + B as on `world` ranks: node k is built in the set of rank k % world, packed there, the records are
    # gathered (rank-major, padded to equal shares) and put into node order, ONE unpack installs them all
    world = 4
    cells_cap = (int(cells_per_map.max()) * 5 // 4 + 63) // 64 * 64
    n_max = (n_nodes + world - 1) // world
    stride = direct.pack_bytes(cells_cap, True)
    assert stride == D.record_bytes(cells_cap, 200 * 200 * 2)
    gathered = torch.zeros((world, n_max, stride), dtype=torch.uint8, device=dev)
    locs = []
    for rank in range(world):
        mine = D.shard_nodes(n_nodes, rank, world)
        loc = N.MapSet(res, [0, 0, 0], size, n_maps=len(mine), max_cells=2048)
        loc.enable_occupancy()
        sel = torch.as_tensor(mine, device=dev)
        for (sc, org), kw in zip(clouds, fuse_kw):
            loc.add_cloud(sc[sel].contiguous(), org[mine], **kw)
        buf = torch.zeros((len(mine), stride), dtype=torch.uint8, device=dev)
        loc.pack_cells(buf, 0, len(mine), cells_cap=cells_cap, with_occupancy=True)
        gathered[rank, :len(mine)] = buf
        torch.cuda.synchronize()
        locs.append(loc)
    allrec = D.records_to_node_order(gathered.view(world * n_max, stride), n_nodes, world)
    pool = N.MapSet(res, [0, 0, 0], size, n_maps=n_nodes, max_cells=2048)
    pool.enable_occupancy()
    pool.unpack_cells(allrec, 0, n_nodes, with_occupancy=True)
    # a record on the host reads like the exported map of the rank that built it
    for k in (0, 137, 499):
        mean, cov, slot, npts, flags = D.cells_from_record(allrec[k].cpu().numpy())
        em, ec, ei, en = locs[k % world].export_cells(k // world)
        assert flags == 2 and np.array_equal(mean, em) and np.array_equal(cov, ec) and np.array_equal(npts, en)
        assert np.array_equal(slot, (ei[:, 0] * 200 + ei[:, 1]) * 2 + ei[:, 2])
    # an unpacked map is the packed map: cells, counters, occupancies (and the matcher's bits: test_pack_... below)
    assert np.array_equal(pool.num_cells_all(), cells_per_map)
    for k in (3, 250, 498, 499):
        for x, y in zip(pool.export_cells(k), locs[k % world].export_cells(k // world)):
            assert np.array_equal(x, y)
        assert np.array_equal(pool.occupancy(k), locs[k % world].occupancy(k // world))
        # against the same node built in the 500-map set: the same cells and point counts; the moments may differ in
        # their last bits (a set of 125 maps cuts a scan into other chunks than a set of 500: other partial sums)
        a, b = pool.export_cells(k), direct.export_cells(k)
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
        assert np.max(np.abs(a[0] - b[0])) < 1e-12 and np.max(np.abs(a[1] - b[1])) < 1e-12

    # the SPARSE form of the same exchange: occupancies as (slot, value) pairs of the cells with a reading, in slot order.
    # Same maps after unpacking (into a set whose occupancies held something else before), a seventh of the bytes.
    occ_cap = (max(loc.occupied_cells_max() for loc in locs) + 63) // 64 * 64
    stride_s = direct.pack_bytes(cells_cap, occ_cap=occ_cap)
    assert stride_s == D.record_bytes(cells_cap, occ_cap=occ_cap) and 5 * stride_s < stride
    gathered_s = torch.zeros((world, n_max, stride_s), dtype=torch.uint8, device=dev)
    for rank, loc in enumerate(locs):
        mine = D.shard_nodes(n_nodes, rank, world)
        loc.pack_cells(gathered_s[rank], 0, len(mine), cells_cap=cells_cap, occ_cap=occ_cap)
        again = torch.zeros((len(mine), stride_s), dtype=torch.uint8, device=dev)
        loc.pack_cells(again, 0, len(mine), cells_cap=cells_cap, occ_cap=occ_cap)
        torch.cuda.synchronize()
        assert torch.equal(again, gathered_s[rank, :len(mine)])            # slot order: the bytes do not depend on timing
    allrec_s = D.records_to_node_order(gathered_s.view(world * n_max, stride_s), n_nodes, world)
    pool_s = N.MapSet(res, [0, 0, 0], size, n_maps=n_nodes, max_cells=2048)
    pool_s.enable_occupancy()
    pool_s.unpack_cells(allrec[torch.arange(n_nodes - 1, -1, -1, device=dev)].contiguous(), 0, n_nodes, with_occupancy=True)   # (other maps' readings first)
    pool_s.unpack_cells(allrec_s, 0, n_nodes, with_occupancy=True)
    assert np.array_equal(pool_s.num_cells_all(), cells_per_map)
    for k in (0, 3, 250, 498, 499):
        for x, y in zip(pool_s.export_cells(k), pool.export_cells(k)):
            assert np.array_equal(x, y)
        occ = pool.occupancy(k)
        assert np.array_equal(pool_s.occupancy(k), occ)
        rec = allrec_s[k].cpu().numpy()
        assert D.cells_from_record(rec)[4] == 6
        sl, val = D.sparse_occupancy_of_record(rec)
        flat = occ.reshape(-1)
        assert np.array_equal(sl, np.flatnonzero(flat)) and np.array_equal(val, flat[sl])
    # a record without room for every reading is cut and flagged, and the unpacked map refused like an overflowing build
    small = torch.zeros((1, direct.pack_bytes(cells_cap, occ_cap=64)), dtype=torch.uint8, device=dev)
    locs[0].pack_cells(small, 0, 1, cells_cap=cells_cap, occ_cap=64)
    torch.cuda.synchronize()
    assert D.cells_from_record(small[0].cpu().numpy())[4] == 7
    pool_s.unpack_cells(small, 0, 1, with_occupancy=True)
    with pytest.raises(N.NdtGpuError):
        pool_s.num_cells(0)
    pool_s.unpack_cells(allrec_s[:1].contiguous(), 0, 1, with_occupancy=True)
    assert pool_s.num_cells(0) == cells_per_map[0]

    # phase C: gated candidates and all pairs, dealt block-cyclically to 8 shards, reassembled in edge order
    edges = D.all_pairs(n_nodes)
    assert len(edges) == 124750
    d_odo = np.linalg.norm(odo_T[edges[:, 0], :2, 3] - odo_T[edges[:, 1], :2, 3], axis=1)
    gate = (d_odo <= 6.0) & ((edges[:, 1] - edges[:, 0]) >= 2)
    T0 = np.einsum("eij,ejk->eik", np.linalg.inv(odo_T)[edges[:, 0]], odo_T[edges[:, 1]])
    T_all = np.zeros((len(edges), 4, 4))
    r_all = None
    for rank in range(8):
        mine = D.shard_edges(len(edges), rank, 8, 256)
        Tm, rm = N.match_batch(pool, edges[mine, 0], pool, edges[mine, 1], T0[mine], delta_score=1e-3)   # "edge" preset
        if r_all is None:
            r_all = np.zeros(len(edges), dtype=rm.dtype)
        T_all[mine], r_all[mine] = Tm, rm
    gi = np.nonzero(gate)[0]
    assert 3000 < len(gi) < 6000
    Tg, rg = N.match_batch(pool, edges[gi, 0], pool, edges[gi, 1], T0[gi], delta_score=1e-3)     # the gated edges on their own
    assert np.array_equal(Tg, T_all[gi])
    for f in DET_FIELDS:
        assert np.array_equal(rg[f], r_all[f][gi]), f
    Td, rd = N.match_batch(direct, edges[gi, 0], direct, edges[gi, 1], T0[gi], delta_score=1e-3)  # on the maps built in one set
    assert np.max(np.abs(Td - Tg)) < 1e-7 and np.mean(rd["iterations"] == rg["iterations"]) > 0.99
    assert r_all["converged"][gi].mean() > 0.98
    gt = np.einsum("eij,ejk->eik", np.linalg.inv(node_T)[edges[gi, 0]], node_T[edges[gi, 1]])
    err = np.linalg.norm(Tg[:, :2, 3] - gt[:, :2, 3], axis=1)
    assert np.median(err) < 0.02
    # edges between rooms share no cell: they end at the first evaluation, untouched
    far = room[edges[:, 0]] != room[edges[:, 1]]
    assert np.all(r_all["pair_terms_h"][far] == 0) and np.array_equal(T_all[far], T0[far])

    # per-link outputs of updateLinksUsingNDTRegistration on the gated edges
    cov, sing = N.covariance(pool, edges[gi, 0], pool, edges[gi, 1], Tg)
    score, nb = N.overlap_score(pool, edges[gi, 0], pool, edges[gi, 1], Tg)
    assert sing.mean() < 0.05            # (a few node maps of this small replay hold a dozen cells: singular Hessians)

    # 300 sampled edges against the oracle: the gated ones of three anchor stretches (60 fused oracle maps), the rest others
    anchors = np.concatenate([np.arange(20, 40), np.arange(150, 170), np.arange(420, 440)])
    in_a = np.isin(edges[:, 0], anchors) & np.isin(edges[:, 1], anchors)
    cand_g = np.nonzero(gate & in_a)[0]
    cand_o = np.nonzero(~gate & in_a)[0]
    n_g = min(240, len(cand_g))
    sample = np.concatenate([g.choice(cand_g, n_g, replace=False), g.choice(cand_o, 300 - n_g, replace=False)])
    assert len(sample) == 300 and n_g >= 150
    scans_h = [(sc[torch.as_tensor(anchors, device=dev)].cpu().numpy(), org[anchors]) for sc, org in clouds]
    omaps = {}
    for a, k in enumerate(anchors):
        om = O.OracleMap(res, [0, 0, 0], size)
        for (sc, org), kw in zip(scans_h, fuse_kw):
            om.add_point_cloud(org[a], sc[a], maxz=kw["maxz"], sensor_noise=kw["sensor_noise"], order_free=True)
            om.compute_cells_full()
        assert om.num_cells() == cells_per_map[k]
        omaps[int(k)] = om
    pos = {int(e): q for q, e in enumerate(gi)}
    for e in sample:
        i, j = (int(v) for v in edges[e])
        To, ro = O.match_d2d(omaps[i], omaps[j], T0[e], delta_score=1e-3)
        dt, dr = pose_close(T_all[e], To)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (e, dt, dr)
        assert r_all["iterations"][e] == ro["iterations"] and bool(r_all["converged"][e]) == ro["converged"], e
        if e in pos:
            q = pos[e]
            if sing[q]:
                continue
            co = O.covariance(omaps[i], omaps[j], Tg[q])
            scale = np.abs(co).max()
            assert scale > 0 and np.max(np.abs(cov[q] - co)) < 1e-6 * scale, (e, np.max(np.abs(cov[q] - co)) / scale)
            so, nbo = O.overlap_score(omaps[i], omaps[j], Tg[q])
            assert nb[q] == nbo and abs(score[q] - so) < 1e-6 * max(so, 1e-3), (e, nb[q], nbo, score[q], so)


def test_pack_flags_a_map_that_does_not_fit(N):
    """A record with room for fewer cells than the map holds is cut and flagged; the unpacked map is refused by the
    matcher like a map whose build overflowed max_cells (exit code -3), never silently matched with missing cells."""
    import torch
    from ndt_feature_graph_amd import distributed as D, synth
    dev = torch.device("cuda", 0)
    pr = synth.pair_2d([5], 20000)
    ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=2, max_cells=1024)
    ms.build(np.stack([pr["fixed"][0].numpy(), pr["moving"][0].numpy()]), range_limit=30.0)
    n0 = ms.num_cells(0)
    assert n0 > 64
    buf = torch.zeros((2, ms.pack_bytes(64, False)), dtype=torch.uint8, device=dev)
    ms.pack_cells(buf, 0, 2, cells_cap=64)
    torch.cuda.synchronize()
    _, _, slot, _, flags = D.cells_from_record(buf[0].cpu().numpy())
    assert flags & 1 and len(slot) == 64
    dst = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=2, max_cells=1024)
    dst.unpack_cells(buf, 0, 2)
    with pytest.raises(N.NdtGpuError):
        dst.num_cells(0)                                   # ERR_CAPACITY, like an overflowing build
    T, r = N.match_batch(dst, [0], dst, [1], pr["T_init"].numpy()[:1])
    assert r["exit_code"][0] == -3 and not r["converged"][0]
    # with room for all cells the round trip is exact
    buf2 = torch.zeros((2, ms.pack_bytes(1024, False)), dtype=torch.uint8, device=dev)
    ms.pack_cells(buf2, 0, 2, cells_cap=1024)
    dst.unpack_cells(buf2, 0, 2)
    for k in range(2):
        for x, y in zip(dst.export_cells(k), ms.export_cells(k)):
            assert np.array_equal(x, y)
    Ta, ra = N.match_batch(ms, [0], ms, [1], pr["T_init"].numpy()[:1])
    Tb, rb = N.match_batch(dst, [0], dst, [1], pr["T_init"].numpy()[:1])
    assert np.array_equal(Ta, Tb) and ra["iterations"][0] == rb["iterations"][0]


def test_host_clouds_through_the_pinned_ring(N, monkeypatch):
    """ndtgpu_mapset_build_host / _add_cloud_host on a batch large enough for the chunked path (pinned ring, worker
    threads, the copy of chunk k + 1 under the build of chunk k; more chunks than ring slots): the same maps as the
    one-copy path and as the device-pointer path (cell sets and point counts exact, moments to rounding: the chunks are
    cut into other launch shapes), also through the asynchronous form on a caller's stream."""
    import torch
    from ndt_feature_graph_amd import synth
    dev = torch.device("cuda", 0)
    B, n_pts = 104, 100000                                 # 125 MB: 8 chunks of 13 clouds through 6 slots
    pr = synth.pair_2d(torch.arange(1, B // 2 + 1, device=dev), n_pts, device=dev)
    scans_d = torch.cat([pr["fixed"], pr["moving"]]).contiguous()
    scans = scans_d.cpu().numpy()

    def cells(ms):
        return [ms.export_cells(k) for k in (0, 12, 13, 51, 103)]

    ref = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=B, max_cells=4096)
    ref.build(scans_d, range_limit=30.0)
    want = cells(ref)
    for mode in ("1", "0"):
        monkeypatch.setenv("NDTGPU_HOST_PIPE", mode)
        ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=B, max_cells=4096)
        ms.build(scans, range_limit=30.0)                  # synchronous host form
        ms.build(scans, range_limit=30.0)                  # again: the ring and the staging area are reused
        st = torch.cuda.Stream(device=dev)
        ms2 = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=B, max_cells=4096)
        ms2.build(scans, range_limit=30.0, stream=st)      # asynchronous host form
        st.synchronize()
        for got in (cells(ms), cells(ms2)):
            for a, b in zip(got, want):
                assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
                assert np.max(np.abs(a[0] - b[0])) < 1e-12 and np.max(np.abs(a[1] - b[1])) < 1e-12
        assert np.array_equal(ms.num_cells_all(), ref.num_cells_all())
    # per-map range origins with the chunked path (they go up once, before the first chunk; every chunk reads its own rows)
    g = np.random.default_rng(3)
    origins = g.uniform(-4.0, 4.0, size=(B, 3)) * np.array([1.0, 1.0, 0.0])
    ref.build(scans_d, range_limit=12.0, range_origins=origins)
    want_o, n_o = cells(ref), ref.num_cells_all()
    for mode in ("1", "0"):
        monkeypatch.setenv("NDTGPU_HOST_PIPE", mode)
        ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=B, max_cells=4096)
        ms.build(scans, range_limit=12.0, range_origins=origins)
        for a, b in zip(cells(ms), want_o):
            assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
        assert np.array_equal(ms.num_cells_all(), n_o)
    # fused node maps from host clouds, chunked against one copy
    org = np.zeros((B, 3))
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("NDTGPU_HOST_PIPE", mode)
        ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=B, max_cells=4096)
        ms.enable_occupancy()
        ms.add_cloud(scans, org, maxz=100.0, sensor_noise=0.1)
        ms.add_cloud(scans, org)
        outs.append((cells(ms), ms.occupancy(51)))
    for a, b in zip(outs[0][0], outs[1][0]):
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.max(np.abs(a[1] - b[1])) < 1e-12
    assert np.array_equal(outs[0][1], outs[1][1])


def test_config4_full_size(N, O):
    """configs[3] at FULL size: 5000 fused node maps (10 ray-traced scans of 20 k points each, bench.py --config 4's node),
    built data-parallel as on 8 ranks, exchanged as packed records, the 44 k gated candidate edges registered in the
    block-cyclic shards of 8 ranks.  Properties over all of them (results independent of the shard they were registered in,
    counters against the maps, convergence, the true relative pose recovered, the reference's getValidLinks keeps them);
    300 sampled edges against the CPU oracle (pose, iterations, convergence)."""
    import torch
    from ndt_feature_graph_amd import binding, distributed as D, synth
    dev = torch.device("cuda", 0)
    n_nodes, S, n_pts, res, size, world = 5000, 10, 20000, 0.5, [100.0, 100.0, 1.0], 8
    room, local, world_pose = replay_layout(n_nodes)
    node_T = synth.pose2d_to_T(torch.as_tensor(world_pose)).numpy()
    g = np.random.default_rng(11)
    odo_T = node_T.copy()
    odo_T[:, 0, 3] += g.normal(scale=0.03, size=n_nodes)
    odo_T[:, 1, 3] += g.normal(scale=0.03, size=n_nodes)
    fuse_kw = [dict(maxz=100.0, sensor_noise=0.1)] + [dict(maxz=25.0, sensor_noise=0.06)] * (S - 1)
    anchors = np.concatenate([np.arange(10, 40), np.arange(1230, 1260), np.arange(2510, 2540), np.arange(4950, 4980)])
    seeds = torch.as_tensor(4000 + room, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream()

    # phase A as on 8 ranks: node k in the set of rank k % 8; the scans of one step at a time (1.2 GB each)
    locs, mines = [], []
    for rank in range(world):
        mine = D.shard_nodes(n_nodes, rank, world)
        loc = N.MapSet(res, [0, 0, 0], size, n_maps=len(mine), max_cells=4096)
        loc.enable_occupancy()
        locs.append(loc); mines.append(mine)
    scans_anchor = []
    for k in range(S):
        dx = 0.9 * k / S
        pk = local.copy()
        pk[:, 0] += dx * np.cos(local[:, 2]); pk[:, 1] += dx * np.sin(local[:, 2])
        sc = synth.scan_2d(seeds, torch.as_tensor(pk, device=dev), n_pts, noise_stream=k, chunk_bytes=2 << 30).contiguous()
        sc[:, :, 0] += dx
        org = np.tile(np.array([[dx, 0.0, 0.0]]), (n_nodes, 1))
        for loc, mine in zip(locs, mines):
            loc.add_cloud(sc[torch.as_tensor(mine, device=dev)].contiguous(), org[mine], stream=st, **fuse_kw[k])
        scans_anchor.append((sc[torch.as_tensor(anchors, device=dev)].cpu().numpy(), org[anchors]))
        torch.cuda.synchronize()
        del sc
    cells_local = [loc.num_cells_all() for loc in locs]
    cap_max = max(int(c.max()) for c in cells_local)
    assert cap_max < 4096 and min(int(c.min()) for c in cells_local) > 5
    # phase B: packed records, rank-major -> node order, ONE unpack
    cells_cap = min(4096, (cap_max * 5 // 4 + 63) // 64 * 64)
    n_max = (n_nodes + world - 1) // world
    occ_cap = (max(loc.occupied_cells_max(stream=st) for loc in locs) * 5 // 4 + 63) // 64 * 64    # (the maximum over ranks)
    stride = locs[0].pack_bytes(cells_cap, occ_cap=occ_cap)
    assert 2 * stride < locs[0].pack_bytes(cells_cap, True)            # ~120 KB per node instead of 370 (sized by the node with the most readings)
    gathered = torch.zeros((world, n_max, stride), dtype=torch.uint8, device=dev)
    for rank, (loc, mine) in enumerate(zip(locs, mines)):
        loc.pack_cells(gathered[rank], 0, len(mine), cells_cap=cells_cap, occ_cap=occ_cap, stream=st)
    allrec = D.records_to_node_order(gathered.view(world * n_max, stride), n_nodes, world)
    pool = N.MapSet(res, [0, 0, 0], size, n_maps=n_nodes, max_cells=4096)
    pool.enable_occupancy()
    pool.unpack_cells(allrec, 0, n_nodes, with_occupancy=True, stream=st)
    torch.cuda.synchronize()
    cells_per_map = pool.num_cells_all()
    for rank in range(world):
        assert np.array_equal(cells_per_map[mines[rank]], cells_local[rank])
    for k in (0, 2517, 4999):
        for x, y in zip(pool.export_cells(k), locs[k % world].export_cells(k // world)):
            assert np.array_equal(x, y)
        assert np.array_equal(pool.occupancy(k), locs[k % world].occupancy(k // world))
    del gathered, allrec

    # phase C: the gated candidate edges (bench.py's gate: odometry poses within 6 m, >= 2 indices apart), in 8 shards
    edges_all = D.all_pairs(n_nodes)
    assert len(edges_all) == 12497500
    d_odo = np.linalg.norm(odo_T[edges_all[:, 0], :2, 3] - odo_T[edges_all[:, 1], :2, 3], axis=1)
    gi_all = np.nonzero((d_odo <= 6.0) & ((edges_all[:, 1] - edges_all[:, 0]) >= 2))[0]
    edges = edges_all[gi_all]
    del edges_all, d_odo
    assert 40000 < len(edges) < 50000
    T0 = np.einsum("eij,ejk->eik", np.linalg.inv(odo_T)[edges[:, 0]], odo_T[edges[:, 1]])
    T_all = np.zeros((len(edges), 4, 4))
    r_all = None
    for rank in range(world):
        mine = D.shard_edges(len(edges), rank, world, 256)
        Tm, rm = N.match_batch(pool, edges[mine, 0], pool, edges[mine, 1], T0[mine], delta_score=1e-3)     # the "edge" preset
        if r_all is None:
            r_all = np.zeros(len(edges), dtype=rm.dtype)
        T_all[mine], r_all[mine] = Tm, rm
    # ... and in ONE call: an edge's result does not depend on its shard or its place in the batch
    Tg, rg = N.match_batch(pool, edges[:, 0], pool, edges[:, 1], T0, delta_score=1e-3)
    assert np.array_equal(Tg, T_all)
    for f in DET_FIELDS:
        assert np.array_equal(rg[f], r_all[f]), f
    assert np.array_equal(r_all["n_target"], cells_per_map[edges[:, 0]]) and np.array_equal(r_all["n_source"], cells_per_map[edges[:, 1]])
    assert np.all(r_all["exit_code"] >= 0) and np.all(np.isfinite(T_all)) and np.all(r_all["iterations"] <= 32)
    assert r_all["converged"].mean() > 0.98
    same_room = room[edges[:, 0]] == room[edges[:, 1]]
    assert same_room.all()                                   # (rooms are 80 m apart: the gate never joins two of them)
    gt = np.einsum("eij,ejk->eik", np.linalg.inv(node_T)[edges[:, 0]], node_T[edges[:, 1]])
    err = np.linalg.norm(T_all[:, :2, 3] - gt[:, :2, 3], axis=1)
    assert np.median(err) < 0.02 and (err < 0.05).mean() > 0.9
    # the reference's filter on registered links (getValidLinks, ndt_feature_graph.cpp:527-556; defaults
    # ndt_feature_graph_opt.cpp:49-52: 1.0 m / 0.2 rad / 2 indices) keeps what was registered onto the truth
    kept = D.valid_links(edges, T_all, odo_T)
    assert len(kept) > 0.95 * len(edges)

    # 300 sampled edges against the oracle, between the 120 anchor nodes (their fused maps built by the CPU ray tracer)
    in_a = np.isin(edges[:, 0], anchors) & np.isin(edges[:, 1], anchors)
    cand = np.nonzero(in_a)[0]
    assert len(cand) >= 300
    sample = g.choice(cand, 300, replace=False)
    omaps = {}
    for a, k in enumerate(anchors):
        om = O.OracleMap(res, [0, 0, 0], size)
        for (sc, org), kw in zip(scans_anchor, fuse_kw):
            om.add_point_cloud(org[a], sc[a], maxz=kw["maxz"], sensor_noise=kw["sensor_noise"], order_free=True)
            om.compute_cells_full()
        assert om.num_cells() == cells_per_map[k], (k, om.num_cells(), cells_per_map[k])
        omaps[int(k)] = om
    for e in sample:
        i, j = (int(v) for v in edges[e])
        To, ro = O.match_d2d(omaps[i], omaps[j], T0[e], delta_score=1e-3)
        dt, dr = pose_close(T_all[e], To)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (e, dt, dr)
        assert r_all["iterations"][e] == ro["iterations"] and bool(r_all["converged"][e]) == ro["converged"], e
