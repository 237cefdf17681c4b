"""HIP path vs CPU oracle through the C-ABI (include/ndtgpu.h).  Needs a real MI355X: -m gpu.

Bars (BASELINE.json north_star): integer / index work bit-exact (cell set, point counts); cell
statistics 1e-9 (fixed-point moments); derivatives 1e-9 relative; final pose within
1e-4 m / 1e-4 rad of the CPU matcher on identical inputs."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4
POSE_TOL_RAD = 1e-4


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
    # angle of Ra^T Rb from the chord ||Ra - Rb||_F = 2 sqrt(2) sin(angle/2): well conditioned near 0
    # (arccos of the trace loses half the digits there)
    return float(2.0 * np.arcsin(min(1.0, np.linalg.norm(Ra - Rb) / (2.0 * np.sqrt(2.0)))))


def pose_close(Ta, Tb):
    return np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]), rot_angle(Ta[:3, :3], Tb[:3, :3])


def oracle_map(O, pts, res, size, centre=(0, 0, 0), rng=30.0, n_min=3):
    m = O.OracleMap(res, centre, size)
    m.load_points(pts, rng)
    m.compute_cells(n_min=n_min)
    return m


def assert_cells_equal(gpu, cpu, res):
    gm, gc, gi, gn = gpu
    cm, cc, ci, cn = cpu
    assert len(gn) == len(cn), "number of Gaussian cells differs: %d vs %d" % (len(gn), len(cn))
    assert np.array_equal(gi, ci), "cell index sets differ"
    assert np.array_equal(gn.astype(np.int64), cn.astype(np.int64)), "per-cell point counts differ"
    assert np.max(np.abs(gm - cm)) < 1e-9 * max(1.0, res)
    scale = np.max(np.abs(cc), axis=(1, 2), keepdims=True)
    assert np.max(np.abs(gc - cc) / scale) < 1e-8


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_pts,res", [(10000, 1.0), (100000, 0.5)])
def test_build_parity_scan(N, O, n_pts, res):
    from ndt_feature_graph_amd import synth
    pr = synth.pair_2d([11, 12], n_pts)
    ms = N.MapSet(res, [0, 0, 0], [100, 100, 1], n_maps=4)
    pts = np.concatenate([pr["fixed"].numpy(), pr["moving"].numpy()])        # 4 scans
    ms.build(pts, range_limit=30.0)
    for k in range(4):
        cpu = oracle_map(O, pts[k], res, [100, 100, 1])
        assert_cells_equal(ms.export_cells(k), cpu.export_cells(), res)


def test_build_device_pointer_path_and_stride16(N, O):
    import torch
    from ndt_feature_graph_amd import synth
    pr = synth.pair_2d([21], 30000)
    p3 = pr["fixed"][0]
    p4 = torch.cat([p3, torch.full((p3.shape[0], 1), 7.0)], dim=1).contiguous()   # pcl::PointXYZ padding
    ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=2)
    ms.build(p3[None].cuda().contiguous(), range_limit=30.0, first=0)
    ms.build(p4[None].cuda().contiguous(), range_limit=30.0, first=1)
    torch.cuda.synchronize()
    cpu = oracle_map(O, p3.numpy(), 0.5, [100, 100, 1]).export_cells()
    assert_cells_equal(ms.export_cells(0), cpu, 0.5)
    assert_cells_equal(ms.export_cells(1), cpu, 0.5)


@pytest.mark.parametrize("n_maps,n_pts,stride_dw,res,centre,size", [
    (300, 5003, 3, 0.5, (0.0, 0.0, 0.0), [100, 100, 1]),          # one workgroup per map, last super-tile ragged
    (260, 4096, 4, 0.4, (3.3, -1.7, 0.1), [60, 60, 1.2]),         # pcl::PointXYZ records, cell origins not fp32 numbers
    (3, 20011, 3, 0.5, (0.0, 0.0, 0.0), [100, 100, 1]),           # a few maps: split accumulate + finalise launches
    (2, 9001, 4, 1.0, (-2.5, 4.0, 0.0), [80, 80, 2]),
    (513, 3000, 3, 0.3, (0.0, 0.0, 0.0), [60, 60, 0.6]),          # more maps than resident workgroups, 0.3 m cells
])
def test_build_launch_shapes_and_strides(N, O, n_maps, n_pts, stride_dw, res, centre, size):
    """Every launch shape (one workgroup per map / several per map + finalise), staged record stride (12 B, 16 B; any
    other stride: test_build_odd_grid_and_generic_stride) and grid kind (fp32 cell origins or not) of the build kernel
    against the oracle, with scans whose length is not a multiple of anything and a few NaN points."""
    import torch
    from ndt_feature_graph_amd import synth
    seeds = [31 + (k % 7) for k in range(n_maps)]
    g = np.random.default_rng(n_maps * 1000 + n_pts)
    poses = np.stack([g.uniform(-1, 1, 7), g.uniform(-1, 1, 7), g.uniform(-3.1, 3.1, 7)], axis=1)
    base = synth.scan_2d([31 + k for k in range(7)], poses, n_pts).numpy()         # 7 distinct scans
    base[:, ::97, 0] = np.nan                                                      # dropped beams
    scans = base[np.arange(n_maps) % 7]
    rec = np.full((n_maps, n_pts, stride_dw), 7.0, dtype=np.float32)
    rec[:, :, :3] = scans
    ms = N.MapSet(res, list(centre), size, n_maps=n_maps, max_cells=4096)
    dev = torch.from_numpy(rec).cuda()
    ms.build(dev, range_limit=25.0)                                                # [B, N, 3] or [B, N, 4] records
    torch.cuda.synchronize()
    ref = [oracle_map(O, base[k], res, size, centre=centre, rng=25.0).export_cells() for k in range(7)]
    for m in sorted(set([0, 1, 6, 7, n_maps // 2, n_maps - 1]) & set(range(n_maps))):
        assert_cells_equal(ms.export_cells(m), ref[m % 7], res)


@pytest.mark.parametrize("n_maps,res,centre", [(4, 0.5, (0.0, 0.0, 0.0)), (264, 0.5, (0.0, 0.0, 0.0)), (4, 0.4, (0.13, -0.07, 0.02)),
                                               (260, 0.4, (0.13, -0.07, 0.02)), (3, 0.2, (0.0, 0.0, 0.0))])
def test_build_3d_sweeps_thick_grid(N, O, n_maps, res, centre):
    """3D sweeps (consecutive points change cell every few points) in a thick grid: replaced runs go through the wide
    flush list, with a few maps (split launches; the 0.2 m grid is big enough for the ranking launch on several
    workgroups per map) and with one workgroup per map, on grids with and without fp32 cell origins."""
    import torch
    from ndt_feature_graph_amd import synth
    pr = synth.pair_3d([1, 2], rings=16, azimuths=700)
    base = np.concatenate([pr["fixed"].numpy(), pr["moving"].numpy()])             # 4 distinct sweeps of 11 200 points
    size, rng = [80.0, 80.0, 10.0], 60.0
    ms = N.MapSet(res, list(centre), size, n_maps=n_maps, max_cells=16384)
    assert ms.info()["cells_per_axis"][2] > 4
    ms.build(torch.from_numpy(base[np.arange(n_maps) % 4]).cuda(), range_limit=rng)
    torch.cuda.synchronize()
    ref = [oracle_map(O, base[k], res, size, centre=centre, rng=rng).export_cells() for k in range(4)]
    assert len(ref[0][3]) > 500
    for m in sorted(set([0, 1, 2, 3, n_maps // 2, n_maps - 1]) & set(range(n_maps))):
        assert ms.counters(m)["overflow"] == 0
        assert_cells_equal(ms.export_cells(m), ref[m % 4], res)


def test_build_golden_cells(N, golden):
    for k in range(6):
        pts = golden["cell%d_pts" % k]
        centre = np.round(pts.mean(axis=0).astype(float) * 2) / 2
        ms = N.MapSet(0.5, centre, [0.5, 0.5, 0.5])
        ms.build(pts[None])
        if not bool(golden["cell%d_ok" % k]):
            assert ms.num_cells() == 0
            continue
        mean, cov, idx, n = ms.export_cells()
        assert len(n) == 1 and n[0] == len(pts)
        np.testing.assert_allclose(mean[0], golden["cell%d_mean" % k], rtol=0, atol=1e-10)
        want = golden["cell%d_cov" % k]
        assert np.max(np.abs(cov[0] - want)) < 1e-8 * np.max(np.abs(want))


def test_build_edge_cases(N, O):
    ms = N.MapSet(1.0, [0, 0, 0], [10, 10, 2], n_maps=3)
    base = np.array([[0.1, 0.1, 0.1], [0.2, 0.15, 0.05], [0.15, 0.3, 0.2], [0.3, 0.2, 0.15], [0.25, 0.1, 0.3]], np.float32)
    junk = np.array([[np.nan, 0, 0], [0, np.nan, 0], [4.0, 0.2, 0.1], [100.0, 0, 0], [0, -5.6, 0]], np.float32)
    allnan = np.full((10, 3), np.nan, np.float32)
    ms.build(np.stack([np.concatenate([base, junk]), np.concatenate([junk, base]), allnan]), range_limit=3.0)
    cpu = oracle_map(O, np.concatenate([base, junk]), 1.0, [10, 10, 2], rng=3.0)
    assert_cells_equal(ms.export_cells(0), cpu.export_cells(), 1.0)
    assert_cells_equal(ms.export_cells(1), cpu.export_cells(), 1.0)        # order of points is irrelevant
    assert ms.num_cells(2) == 0                                           # empty scan
    # rebuild in place replaces the content
    ms.build(allnan[None], first=0)
    assert ms.num_cells(0) == 0
    # range measured from an origin (loadPointCloudCentroid)
    ms.build(np.concatenate([base, junk])[None], range_limit=3.0, range_origins=[[3.0, 0, 0]], first=0)
    m = O.OracleMap(1.0, [0, 0, 0], [10, 10, 2])
    m.load_points(np.concatenate([base, junk]), 3.0, [3.0, 0, 0]); m.compute_cells()
    assert_cells_equal(ms.export_cells(0), m.export_cells(), 1.0)


@pytest.mark.parametrize("centre,rng,res", [((0.0, 0.0, 0.0), 20.0, 0.5), ((12.3, -7.65, 0.2), 0.0, 0.5),
                                            ((-333.3, 200.05, 1.0), 0.0, 0.5), ((-333.3, 200.05, 1.0), 0.0, 0.3)])
def test_build_binning_at_cell_faces(N, O, centre, rng, res):
    """Binning must be bit-identical to floor((p - c)/res + 0.5) + size/2 in fp64 (reference LazyGrid).  The kernel
    evaluates it in fp32 and falls back to the fp64 formula inside a derived guard band around every cell face
    (csrc/ndt_build.hip: face_guard): points ON faces, 1..4 fp32 ulps on either side of them, and just inside /
    outside that band, on an even grid whose centre is not a multiple of the cell size, far from the origin, with a
    cell size whose reciprocal is not an fp32 number (the case where the fp32 evaluation really is off by ~1e-4
    cell: with the guard disabled this test fails), and next to the range sphere."""
    g = np.random.default_rng(7)
    size_cells = np.array([200, 200, 2])
    c = np.array(centre)
    n = 60000
    pts = c + g.uniform(-32.0, 32.0, (n, 3)) * res              # 64 x 64 cells, ~15 points each
    pts[:, 2] = c[2] + g.uniform(-0.4, 0.4, n) * res
    k = 24000
    q = pts[:k].copy()
    axis = g.integers(0, 2, k)                                  # put one coordinate on / next to a cell face
    face = c[axis] + (g.integers(-31, 32, k) - 0.5) * res       # faces are at index-space half integers
    f32 = face.astype(np.float32)
    steps = g.integers(-4, 5, k)
    for _ in range(4):
        f32 = np.where(steps > 0, np.nextafter(f32, np.float32(np.inf)), f32); steps = steps - (steps > 0)
    steps = -np.minimum(g.integers(-4, 5, k), 0)
    for _ in range(4):
        f32 = np.where(steps > 0, np.nextafter(f32, np.float32(-np.inf)), f32); steps = steps - (steps > 0)
    band = g.choice([0.0, 5e-5, -5e-5, 2e-4, -2e-4, 9e-4, -9e-4], k) * res
    q[np.arange(k), axis] = f32.astype(np.float64) + band
    cloud = np.concatenate([q, pts[k:]]).astype(np.float32)
    if rng > 0:                                                  # a ring of points hugging the range sphere
        th = g.uniform(0, 2 * np.pi, 4000)
        rr = rng * (1 + g.choice([0.0, 1e-7, -1e-7, 3e-4, -3e-4, 2e-3, -2e-3], 4000))
        ring = np.stack([rr * np.cos(th), rr * np.sin(th), np.zeros(4000)], axis=1)
        cloud = np.concatenate([cloud, ring.astype(np.float32)])
    ms = N.MapSet(res, list(c), list(size_cells * res))
    ms.build(cloud[None], range_limit=rng)
    cpu = oracle_map(O, cloud, res, list(size_cells * res), centre=tuple(c), rng=rng)
    assert cpu.num_cells() > 2000
    assert_cells_equal(ms.export_cells(0), cpu.export_cells(), res)


def test_build_odd_grid_and_generic_stride(N, O):
    """Odd cell counts (the reference's double->int truncation quirk: every point takes the exact fp64
    index path) and a record stride that is neither 12 nor 16 bytes (generic load path)."""
    import torch
    from ndt_feature_graph_amd import synth
    pts = synth.pair_2d([51], 30000)["fixed"][0].numpy()
    # 33 x 27 x 1 cells of 1.0 m around a centre that is not a multiple of the cell size
    ms = N.MapSet(1.0, [0.37, -1.21, 0.0], [33, 27, 1])
    ms.build(pts[None], range_limit=30.0)
    cpu = oracle_map(O, pts, 1.0, [33, 27, 1], centre=(0.37, -1.21, 0.0))
    assert_cells_equal(ms.export_cells(0), cpu.export_cells(), 1.0)
    # stride 20 bytes: xyz + two floats of payload
    p5 = torch.cat([torch.from_numpy(pts), torch.full((len(pts), 2), 3.0)], dim=1).contiguous().cuda()
    ms2 = N.MapSet(0.5, [0, 0, 0], [100, 100, 1])
    from ndt_feature_graph_amd import binding
    import ctypes as C
    cp = binding.CellParams(3, 1000.0)
    binding._check(binding.lib().ndtgpu_mapset_build(ms2.h, 0, 1, C.c_void_p(p5.data_ptr()), len(pts), 20, 20 * len(pts),
                                                      30.0, None, C.byref(cp), None))
    torch.cuda.synchronize()
    cpu2 = oracle_map(O, pts, 0.5, [100, 100, 1])
    assert_cells_equal(ms2.export_cells(0), cpu2.export_cells(), 0.5)


def test_build_unordered_points_same_result(N):
    """Shuffled input: same cells, same counts (integer work: bit-exact); moments agree to fp64 rounding
    (per-lane partial sums are formed in input order, the cross-wave adds are exact)."""
    from ndt_feature_graph_amd import synth
    pts = synth.pair_2d([31], 50000)["fixed"][0].numpy()
    perm = np.random.default_rng(0).permutation(len(pts))
    ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=3)
    ms.build(np.stack([pts, pts[perm], pts]), range_limit=30.0)
    a, b, c = ms.export_cells(0), ms.export_cells(1), ms.export_cells(2)
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert np.max(np.abs(a[0] - b[0])) < 1e-12
    assert np.max(np.abs(a[1] - b[1]) / np.max(np.abs(a[1]), axis=(1, 2), keepdims=True)) < 1e-10
    for x, y in zip(a, c):
        assert np.array_equal(x, y)          # same input order: bit-identical, whatever the atomic order


def test_capacity_overflow_is_reported(N):
    from ndt_feature_graph_amd import synth
    pts = synth.pair_2d([41], 20000)["fixed"][0].numpy()
    ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], max_cells=16)
    ms.build(pts[None], range_limit=30.0)
    with pytest.raises(N.NdtGpuError) as e:
        ms.num_cells()
    assert e.value.status == -4
    # ... also when a big grid is ranked by several workgroups per map (split launches), and the set stays usable
    p3 = synth.pair_3d([1], rings=16, azimuths=700)["fixed"].numpy()
    big = N.MapSet(0.2, [0, 0, 0], [80, 80, 10], n_maps=2, max_cells=300)
    big.build(np.concatenate([p3, p3]), range_limit=60.0)
    with pytest.raises(N.NdtGpuError) as e:
        big.num_cells(1)
    assert e.value.status == -4


def test_big_grid_rebuilds_leave_a_clean_state(N, O):
    """A grid big enough for the split finalise (rank-map launch + placement launch): the same set is built three times with
    different sweeps -- the second one overflows max_cells and takes the general ranking path -- and the third build must
    equal the oracle: work table, bitmap, accumulators and tickets are back in their clean state after either path."""
    import torch
    from ndt_feature_graph_amd import synth
    pr = synth.pair_3d([3, 4], rings=16, azimuths=700)
    a, b = pr["fixed"].numpy(), pr["moving"].numpy()                    # 2 + 2 sweeps of 11 200 points
    dense = synth.pair_3d([5], rings=64, azimuths=3125)["fixed"].numpy()  # 200 000 points: more cells than the set holds
    res, size, rng = 0.2, [80.0, 80.0, 10.0], 60.0
    ms = N.MapSet(res, [0, 0, 0], size, n_maps=2, max_cells=8000)       # (accumulators: every TOUCHED cell needs one)
    n0 = len(oracle_map(O, a[0], res, size, rng=rng).export_cells()[3])
    assert 500 < n0 < 8000
    ms.build(torch.from_numpy(a).cuda(), range_limit=rng)
    torch.cuda.synchronize()
    for m in range(2):
        assert ms.counters(m)["overflow"] == 0 and ms.counters(m)["n_alloc"] == 0, ms.counters(m)
        assert_cells_equal(ms.export_cells(m), oracle_map(O, a[m], res, size, rng=rng).export_cells(), res)
    ms.build(torch.from_numpy(np.concatenate([dense, dense])).cuda(), range_limit=rng)
    torch.cuda.synchronize()
    assert ms.counters(0)["overflow"] != 0 and ms.counters(1)["overflow"] != 0, ms.counters(0)
    ms.build(torch.from_numpy(b).cuda(), range_limit=rng)
    torch.cuda.synchronize()
    for m in range(2):
        assert ms.counters(m)["overflow"] == 0 and ms.counters(m)["n_alloc"] == 0, ms.counters(m)
        assert_cells_equal(ms.export_cells(m), oracle_map(O, b[m], res, size, rng=rng).export_cells(), res)


# ---------------------------------------------------------------------------------------------
def test_derivatives_golden_and_oracle(N, O, golden):
    res = float(golden["d2d_res"])
    size = golden["d2d_size"] * res
    tg = N.MapSet(res, golden["d2d_centre"], size)
    tg.set_cells(0, golden["d2d_tgt_mean"], golden["d2d_tgt_cov"])
    assert tg.num_cells() == len(golden["d2d_tgt_mean"])
    ot = O.OracleMap(res, golden["d2d_centre"], size)
    ot.set_cells(golden["d2d_tgt_mean"], golden["d2d_tgt_cov"])
    for nn in (0, 1, 2):
        s, g, H = N.derivatives(tg, 0, golden["d2d_src_mean"], golden["d2d_src_cov"], n_neighbours=nn)
        so, go, Ho = O.derivatives(ot, golden["d2d_src_mean"], golden["d2d_src_cov"], n_neighbours=nn)
        assert abs(s - so) < 1e-11 * abs(so)
        assert np.max(np.abs(g - go)) < 1e-10 * np.max(np.abs(go))
        assert np.max(np.abs(H - Ho)) < 1e-10 * np.max(np.abs(Ho))
        s2, g2, _ = N.derivatives(tg, 0, golden["d2d_src_mean"], golden["d2d_src_cov"], n_neighbours=nn,
                                  compute_hessian=False)
        assert abs(s2 - s) < 1e-13 * abs(s) and np.max(np.abs(g2 - g)) < 1e-12 * np.max(np.abs(g))
    s, g, H = N.derivatives(tg, 0, golden["d2d_src_mean"], golden["d2d_src_cov"], n_neighbours=2)
    assert abs(s - float(golden["d2d_score"])) < 1e-11 * abs(s)                 # NumPy restatement
    assert np.max(np.abs(g - golden["d2d_grad_fd"])) < 1e-6 * np.max(np.abs(g))  # vs NumPy finite differences
    assert np.max(np.abs(H - golden["d2d_hess_fd"])) < 2e-5 * np.max(np.abs(H))


@pytest.mark.parametrize("size_cells,nn", [((12, 9, 7), 1), ((12, 9, 7), 2), ((12, 9, 7), 3), ((40, 33, 2), 2),
                                           ((40, 33, 1), 3), ((5, 70, 3), 2), ((6, 6, 6), 0)])
def test_probe_rank_bitmap_windows(N, O, size_cells, nn):
    """The matcher finds the (2n+1)^3 neighbours through bit windows of the rank bitmap (csrc/ndt_match.hip PROBE):
    random lattices of target cells (about half of the slots filled, so windows start at every bit offset and
    straddle 32-slot words), source cells everywhere including outside the grid and on its border; both the
    flat-map form (sz <= n+1: one window per x) and the general form (one window per (x, y)).  The pair set must
    equal getCellsForPoint's: score, gradient and Hessian agree with the oracle to rounding."""
    rng = np.random.default_rng(size_cells[0] * 100 + size_cells[2] * 10 + nn)
    res = 0.5
    size_m = [c * res for c in size_cells]
    sx, sy, sz = size_cells
    ix, iy, iz = np.meshgrid(np.arange(sx), np.arange(sy), np.arange(sz), indexing="ij")
    keep = rng.random(ix.shape) < 0.5
    idx = np.stack([ix[keep], iy[keep], iz[keep]], axis=1)
    # LazyGrid: i = (int)(floor((p - c)/res + 0.5) + size/2.0)  =>  cell i is centred at c + (i - floor(size/2)) * res
    ctr = (idx - np.array(size_cells) // 2) * res
    mean = ctr + rng.uniform(-0.2, 0.2, ctr.shape) * res
    A = rng.normal(size=(len(idx), 3, 3)) * 0.1
    cov = A @ A.transpose(0, 2, 1) + 0.01 * np.eye(3)
    tg = N.MapSet(res, [0, 0, 0], size_m, max_cells=max(4096, len(idx) + 8))
    tg.set_cells(0, mean, cov)
    assert tg.num_cells() == len(idx)
    ot = O.OracleMap(res, [0, 0, 0], size_m)
    ot.set_cells(mean, cov)
    m = 700
    half = np.array(size_m) / 2.0
    smean = rng.uniform(-half - 1.2 * res, half + 1.2 * res, (m, 3))      # some sources outside the grid
    smean[:40] = np.clip(smean[:40], -half + 1e-3, half - 1e-3)
    B = rng.normal(size=(m, 3, 3)) * 0.1
    scov = B @ B.transpose(0, 2, 1) + 0.01 * np.eye(3)
    s, g, H = N.derivatives(tg, 0, smean, scov, n_neighbours=nn)
    so, go, Ho = O.derivatives(ot, smean, scov, n_neighbours=nn)
    assert so != 0.0
    assert abs(s - so) < 1e-10 * abs(so)
    assert np.max(np.abs(g - go)) < 1e-9 * np.max(np.abs(go))
    assert np.max(np.abs(H - Ho)) < 1e-9 * np.max(np.abs(Ho))


def test_derivatives_on_scan_maps(N, O):
    from ndt_feature_graph_amd import synth
    pr = synth.pair_2d([5], 60000)
    f, m = pr["fixed"][0].numpy(), pr["moving"][0].numpy()
    ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=2)
    ms.build(np.stack([f, m]), range_limit=30.0)
    ot = oracle_map(O, f, 0.5, [100, 100, 1])
    mean, cov, _, _ = ms.export_cells(1)
    T = pr["T_init"][0].numpy()
    mm, cc = mean @ T[:3, :3].T + T[:3, 3], T[:3, :3] @ cov @ T[:3, :3].T
    s, g, H = N.derivatives(ms, 0, mm, cc)
    so, go, Ho = O.derivatives(ot, mm, cc)
    assert abs(s - so) < 1e-9 * abs(so)
    assert np.max(np.abs(g - go)) < 1e-9 * np.max(np.abs(go))
    assert np.max(np.abs(H - Ho)) < 1e-9 * np.max(np.abs(Ho))


# ---------------------------------------------------------------------------------------------
def _pair_maps(N, O, seeds, n_pts, res, oracle_maps=True):
    from ndt_feature_graph_amd import synth
    pr = synth.pair_2d(seeds, n_pts)
    B = len(seeds)
    tg = N.MapSet(res, [0, 0, 0], [100, 100, 1], n_maps=B)
    sr = N.MapSet(res, [0, 0, 0], [100, 100, 1], n_maps=B)
    tg.build(pr["fixed"].numpy(), range_limit=30.0)
    sr.build(pr["moving"].numpy(), range_limit=30.0)
    om = [(oracle_map(O, pr["fixed"][b].numpy(), res, [100, 100, 1]),
           oracle_map(O, pr["moving"][b].numpy(), res, [100, 100, 1])) for b in range(B)] if oracle_maps else None
    return pr, tg, sr, om


@pytest.mark.parametrize("n_pts,res,seeds", [(10000, 1.0, [1, 2, 3, 4]), (100000, 0.5, [1, 2, 3, 4, 5, 6])])
def test_match_parity_6dof(N, O, n_pts, res, seeds):
    """configs[0] (10 k pts, 1.0 m) and configs[1] (100 k pts, 0.5 m): pose parity per pair."""
    pr, tg, sr, om = _pair_maps(N, O, seeds, n_pts, res)
    B = len(seeds)
    T0 = pr["T_init"].numpy()
    T, r = N.match_batch(tg, np.arange(B), sr, np.arange(B), T0)
    for b in range(B):
        To, ro = O.match_d2d(om[b][0], om[b][1], T0[b])
        dt, dr = pose_close(T[b], To)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (b, dt, dr)
        assert bool(r["converged"][b]) == ro["converged"]
        assert abs(r["score"][b] - ro["score"]) < 1e-6 * abs(ro["score"])
        assert r["n_source"][b] == om[b][1].num_cells() and r["n_target"][b] == om[b][0].num_cells()
        gt = pr["T_gt"][b].numpy()
        assert pose_close(T[b], gt)[0] < 0.03          # and it is the right answer, not just the same one


def test_match_parity_edge_preset_and_3dof(N, O):
    pr, tg, sr, om = _pair_maps(N, O, [7, 8], 40000, 0.5)
    T0 = pr["T_init"].numpy()
    # "edge" preset: default-constructed NDTMatcherD2D, DELTA_SCORE 1e-3 (graph.cpp:261-262)
    T, r = N.match_batch(tg, [0, 1], sr, [0, 1], T0, delta_score=1e-3)
    for b in range(2):
        To, ro = O.match_d2d(om[b][0], om[b][1], T0[b], delta_score=1e-3)
        dt, dr = pose_close(T[b], To)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD
        assert r["iterations"][b] == ro["iterations"]
    # NDTMatcherD2D_2D: {x, y, yaw}; start near the optimum (the 3x3 regulariser is fragile far away)
    Tn = pr["T_gt"].numpy().copy()
    Tn[:, 0, 3] += 0.02
    Tn[:, 1, 3] -= 0.01
    T, r = N.match_batch(tg, [0, 1], sr, [0, 1], Tn, dof_mask=0x23)
    for b in range(2):
        To, ro = O.match_d2d(om[b][0], om[b][1], Tn[b], dof_mask=0x23)
        dt, dr = pose_close(T[b], To)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD
        assert abs(T[b][2, 3]) < 1e-15 and abs(T[b][2, 2] - 1) < 1e-15       # z, roll, pitch untouched
    # no step control, iteration cap, no initial guess
    T, r = N.match_batch(tg, [0], sr, [0], T0[:1], step_control=0)
    To, ro = O.match_d2d(om[0][0], om[0][1], T0[0], step_control=0)
    assert pose_close(T[0], To)[0] <= POSE_TOL_M
    T, r = N.match_batch(tg, [0], sr, [0], T0[:1], itr_max=1)
    To, ro = O.match_d2d(om[0][0], om[0][1], T0[0], itr_max=1)
    assert (not r["converged"][0]) and r["iterations"][0] == ro["iterations"] == 3 and r["exit_code"][0] == 3
    assert pose_close(T[0], To)[0] <= POSE_TOL_M
    T, r = N.match_batch(tg, [0], sr, [0], T0[:1], use_initial_guess=0)
    To, ro = O.match_d2d(om[0][0], om[0][1], T0[0], use_initial_guess=0)
    dt, dr = pose_close(T[0], To)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD


def test_match_batch_is_deterministic_and_order_free(N, O):
    pr, tg, sr, om = _pair_maps(N, O, [1, 2, 3, 4, 5], 20000, 0.5)
    T0 = pr["T_init"].numpy()
    idx = np.arange(5)
    Ta, ra = N.match_batch(tg, idx, sr, idx, T0)
    Tb, rb = N.match_batch(tg, idx, sr, idx, T0)
    det = ["converged", "iterations", "fevals", "exit_code", "score", "n_source", "n_target"]   # not the cycle counters
    assert np.array_equal(Ta, Tb) and all(np.array_equal(ra[k], rb[k]) for k in det)   # run-to-run identical
    perm = np.array([3, 0, 4, 1, 2])
    Tc, rc = N.match_batch(tg, idx[perm], sr, idx[perm], T0[perm])
    assert np.array_equal(Tc, Ta[perm])                                        # a pair's result ignores its batch
    T1, r1 = N.match_d2d(tg, 2, sr, 2, T0[2])
    assert np.array_equal(T1, Ta[2])


def test_persistent_cooperative_and_host_driven_paths_agree(N, O, monkeypatch):
    """Batches that fill the chip: persistent workgroups pulling pairs; smaller ones: one cooperative launch with
    several workgroups per registration (grid barrier, workgroup 0 solves), or with NDTGPU_HOST_LOOP=1 (<= 8 pairs)
    the host-driven Newton loop + one derivative launch per evaluation.  Same state machine
    (csrc/ndt_solver.h), different summation order."""
    pr, tg, sr, om = _pair_maps(N, O, list(range(1, 13)), 20000, 0.5)
    T0 = pr["T_init"].numpy()
    idx = np.arange(12)
    monkeypatch.setenv("NDTGPU_COOP", "0")
    Tb, rb = N.match_batch(tg, idx, sr, idx, T0)                 # persistent kernel
    monkeypatch.setenv("NDTGPU_COOP", "1")
    Tc, rc = N.match_batch(tg, idx, sr, idx, T0)                 # one grid-barrier launch, a few workgroups per pair
    monkeypatch.delenv("NDTGPU_COOP")
    # (another summation order: agreement to rounding of the sums, far inside the 1e-4 m / rad of the metric)
    assert np.max(np.abs(Tc - Tb)) < 1e-8 and np.array_equal(rc["iterations"], rb["iterations"])
    for b in (0, 5, 11):
        Ts, rs = N.match_d2d(tg, b, sr, b, T0[b])                # cooperative kernel, whole chip
        monkeypatch.setenv("NDTGPU_HOST_LOOP", "1")
        Th, rh = N.match_d2d(tg, b, sr, b, T0[b])                # host-driven loop
        monkeypatch.delenv("NDTGPU_HOST_LOOP")
        for Tx, rx in ((Ts, rs), (Th, rh)):
            dt, dr = pose_close(Tx, Tb[b])
            assert dt < 1e-8 and dr < 1e-8
            assert rx["iterations"] == rb["iterations"][b] and rx["converged"] == rb["converged"][b]
        monkeypatch.setenv("NDTGPU_COOP_CELLS", "16")            # many workgroups: another partition, same answer
        T3, r3 = N.match_d2d(tg, b, sr, b, T0[b])
        monkeypatch.delenv("NDTGPU_COOP_CELLS")
        dt, dr = pose_close(T3, Tb[b])
        assert dt < 1e-8 and dr < 1e-8 and abs(int(r3["fevals"]) - int(rs["fevals"])) <= 2   # a line-search trial more or less
        To, ro = O.match_d2d(om[b][0], om[b][1], T0[b])
        dt, dr = pose_close(Ts, To)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD


def test_grid_barrier_calls_leave_their_control_blocks_clean(N, O, monkeypatch):
    """The grid-barrier matcher (small batches) zeroes its barrier counters when the last workgroup leaves, and the
    host only clears blocks it has not seen finish cleanly at the same stride: calls of different sizes and shapes on
    one map set, in any order, must each give the bits of their first run (a stale counter would stall or skip a
    barrier); with the runtime-checked cooperative launch (NDTGPU_COOP_API=1) the same bits."""
    pr, tg, sr, om = _pair_maps(N, O, list(range(1, 9)), 20000, 0.5, oracle_maps=False)
    T0 = pr["T_init"].numpy()
    first = {}
    seq = [(0,), (0, 1, 2, 3, 4), (3,), tuple(range(8)), (0,), (5, 6), tuple(range(8)), (3,), (0, 1, 2, 3, 4), (5, 6)]
    for rep, which in enumerate(seq + seq):
        if rep == len(seq):
            monkeypatch.setenv("NDTGPU_COOP_API", "1")
        if rep == len(seq) + 4:
            monkeypatch.setenv("NDTGPU_COOP_CELLS", "32")       # another stride on the same set ...
        if rep == len(seq) + 6:
            monkeypatch.delenv("NDTGPU_COOP_CELLS")              # ... and back
        idx = np.array(which)
        T, r = N.match_batch(tg, idx, sr, idx, T0[idx])
        if os.environ.get("NDTGPU_COOP_CELLS"):
            continue
        if which in first:
            assert np.array_equal(T, first[which][0]) and np.array_equal(r["iterations"], first[which][1])
        else:
            first[which] = (T.copy(), r["iterations"].copy())
    for k in range(5):                                           # a pair's result ignores the batch it is in
        assert np.array_equal(first[(0, 1, 2, 3, 4)][0][k], first[tuple(range(8))][0][k])


def test_grid_barrier_result_does_not_depend_on_the_batch(N, O, monkeypatch):
    """The grid-barrier matcher cuts the source cells of a registration into chunks of 128 cells -- a property of the map --
    and adds the chunks' sums in chunk order.  Alone on the chip a registration has one workgroup per chunk; in a batch
    of 72 a workgroup takes four chunks at a time through one pass of 64 cells per wave (eval_chunks), each chunk still
    summed on its own: the bits must not change.  One pair against the oracle."""
    monkeypatch.setenv("NDTGPU_COOP", "1")      # (sets of small maps would take the persistent kernel beyond 8 pairs)
    pr, tg, sr, om = _pair_maps(N, O, list(range(1, 7)), 60000, 0.25)
    T0 = pr["T_init"].numpy()
    assert min(sr.num_cells(k) for k in range(6)) > 4 * 128            # more chunks than workgroups per pair below
    idx = np.arange(72) % 6
    Tb, rb = N.match_batch(tg, idx, sr, idx, T0[idx])                  # 72 pairs: four chunks per task
    for k in range(6):
        T1, r1 = N.match_d2d(tg, k, sr, k, T0[k])                      # alone: a workgroup per chunk
        for j in range(k, 72, 6):
            assert np.array_equal(Tb[j], T1) and rb["iterations"][j] == r1["iterations"] and rb["fevals"][j] == r1["fevals"]
    Td, rd = N.match_batch(tg, idx[:12], sr, idx[:12], T0[idx[:12]])   # 12 pairs: one chunk per task
    assert np.array_equal(Td, Tb[:12])
    # batches of more than 8 pairs take the task pool (any workgroup, any task), up to 8 the grid-barrier kernel (static
    # teams): the same chunks added in the same order -- forced either way, the same bits
    for pool in ("0", "1"):
        monkeypatch.setenv("NDTGPU_POOL", pool)
        Tp, rp = N.match_batch(tg, idx, sr, idx, T0[idx])
        assert np.array_equal(Tp, Tb) and np.array_equal(rp["fevals"], rb["fevals"])
        T1, r1 = N.match_d2d(tg, 3, sr, 3, T0[3])
        assert np.array_equal(T1, Tb[3])
    monkeypatch.delenv("NDTGPU_POOL")
    To, ro = O.match_d2d(om[2][0], om[2][1], T0[2])
    dt, dr = pose_close(Tb[2], To)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD and rb["iterations"][2] == ro["iterations"]


def test_one_and_two_registrations_per_workgroup_agree(N, O, monkeypatch):
    """The persistent matcher keeps two registrations in flight per workgroup and its waves take the eight shares of
    their evaluations in whatever order they come free (NDTGPU_SLOTS=1: one registration per workgroup).  Shares,
    their hit lists and their partial sums are the same whoever computes them and are added in share order, so a
    registration's result does not depend on the form or on timing -- bit for bit; parking and the rule that a
    workgroup resumes one parked registration at a time (NDTGPU_DOUBLE_THRESH) are bit neutral too."""
    n = 700                                                      # > 512 slots: tickets are drawn twice
    pr, tg, sr, om = _pair_maps(N, O, [1 + (k % 24) for k in range(24)], 6000, 1.0, oracle_maps=False)
    idx = np.arange(n) % 24
    T0 = pr["T_init"].numpy()[idx]
    monkeypatch.setenv("NDTGPU_COOP", "0")
    monkeypatch.setenv("NDTGPU_SLOTS", "1")
    Tw, rw = N.match_batch(tg, idx, sr, idx, T0)
    monkeypatch.setenv("NDTGPU_SLOTS", "2")
    Tn, rn = N.match_batch(tg, idx, sr, idx, T0)
    Tn2, rn2 = N.match_batch(tg, idx, sr, idx, T0)
    assert np.array_equal(Tn, Tn2) and np.array_equal(rn["score"], rn2["score"])      # run to run
    monkeypatch.setenv("NDTGPU_PARK_ITERS", "1")
    Tp, rp = N.match_batch(tg, idx, sr, idx, T0)
    assert np.array_equal(Tn, Tp) and np.array_equal(rn["fevals"], rp["fevals"])      # parking is bit neutral
    monkeypatch.setenv("NDTGPU_DOUBLE_THRESH", "0")
    Tq, rq = N.match_batch(tg, idx, sr, idx, T0)
    assert np.array_equal(Tn, Tq) and np.array_equal(rn["fevals"], rq["fevals"])
    assert np.array_equal(Tn, Tw)
    for f in ("converged", "iterations", "fevals", "score", "pair_terms_g", "pair_terms_h"):
        assert np.array_equal(rn[f], rw[f]), f


def test_scheduler_parking_is_bit_neutral(N, O, monkeypatch):
    """The persistent matcher pulls pairs from a ticket counter and parks long registrations
    (csrc/ndt_match.hip: NdtMatchWork).  Parking saves and restores the solver state bit for bit: more pairs than
    CUs, parking after 0 (off) / 1 / 3 iterations -> identical transforms, iteration and evaluation counts."""
    n = 600                                                      # > 256 workgroups: tickets are drawn twice
    pr, tg, sr, om = _pair_maps(N, O, [1 + (k % 24) for k in range(24)], 6000, 1.0, oracle_maps=False)
    idx = np.arange(n) % 24
    T0 = pr["T_init"].numpy()[idx]
    ref = None
    for park in ("0", "1", "3"):
        monkeypatch.setenv("NDTGPU_PARK_ITERS", park)
        T, r = N.match_batch(tg, idx, sr, idx, T0)
        if ref is None:
            ref = (T, r)
            # the same pair gives the same bits wherever it sits in the batch
            assert np.array_equal(T[:24], T[24:48]) and np.array_equal(r["fevals"][:24], r["fevals"][576:600])
        else:
            assert np.array_equal(T, ref[0])
            for f in ("iterations", "fevals", "converged", "exit_code", "score", "pair_terms_g", "pair_terms_h"):
                assert np.array_equal(r[f], ref[1][f]), f


@pytest.mark.parametrize("res,n_pts,nn,step_control", [(0.3, 30000, 1, 1), (1.0, 8000, 3, 1), (0.25, 40000, 2, 1),
                                                        (0.5, 20000, 2, 0)])
def test_match_parity_sweep(N, O, res, n_pts, nn, step_control):
    """Other cell sizes, neighbourhood radii 1 and 3, Newton without step control: 10 pairs each, through the
    cooperative launch (10 pairs) and through the persistent kernel (the same pairs repeated to 200)."""
    seeds = list(range(500, 510))
    B = len(seeds)
    pr, tg, sr, om = _pair_maps(N, O, seeds, n_pts, res)
    T0 = pr["T_init"].numpy()
    kw = dict(n_neighbours=nn, step_control=step_control)
    Tc, rc = N.match_batch(tg, np.arange(B), sr, np.arange(B), T0, **kw)
    rep = np.arange(200) % B
    Tp, rp = N.match_batch(tg, rep, sr, rep, T0[rep], **kw)
    for b in range(B):
        To, ro = O.match_d2d(om[b][0], om[b][1], T0[b], **kw)
        for Tx, rx in ((Tc[b], rc), (Tp[b], rp)):
            dt, dr = pose_close(Tx, To)
            assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (b, dt, dr)
            same_flow = bool(rx["converged"][b]) == ro["converged"] and rx["iterations"][b] == ro["iterations"]
            if not same_flow:
                # One iteration more or less at the convergence test |dp| < DELTA_SCORE is accepted only where the ORACLE ALONE
                # does the same when its own sums are taken in another order or moved by a few ulp (oracle_set_sum_mode, as in
                # test_gpu_fullsize.py): rounding decides there, not the formulas.
                assert ro["iterations"] > 1 and abs(int(rx["iterations"][b]) - ro["iterations"]) <= 1, (b, rx["iterations"][b], ro["iterations"])
                base, fragile = (ro["iterations"], ro["converged"]), False
                for mode in list(range(16, 28)) + list(range(1, 16)):
                    O.set_sum_mode(mode)
                    try:
                        Tq, rq = O.match_d2d(om[b][0], om[b][1], T0[b], **kw)
                    finally:
                        O.set_sum_mode(0)
                    if (rq["iterations"], rq["converged"]) != base:
                        fragile = True
                        break
                assert fragile, (b, rx["iterations"][b], ro["iterations"], "the oracle's flow is stable under other summation orders and ulp noise")


def test_full_size_properties(N):
    """configs[1] size (100 k points, 0.5 m cells), properties that need no oracle:
    (a) grid translation equivariance: both scans and the grid centre moved by a whole number of cells give the same
        cells (shifted) and the same registration; (b) a map matched against itself stays at the identity;
    (c) every point is either binned or counted as dropped; (d) swapping the roles of the scans gives the inverse
        transform up to the matcher's own convergence tolerance."""
    from ndt_feature_graph_amd import synth
    B, n = 8, 100000
    pr = synth.pair_2d(list(range(101, 101 + B)), n)
    f, m, T0 = pr["fixed"].numpy(), pr["moving"].numpy(), pr["T_init"].numpy()
    size = [100, 100, 1]
    a = N.MapSet(0.5, [0, 0, 0], size, n_maps=2 * B)
    a.build(np.concatenate([f, m]), range_limit=30.0)
    idx = np.arange(B)
    Ta, ra = N.match_batch(a, idx, a, idx + B, T0)
    # (a) shift by (+37, -12, 0) cells; the range limit is measured from the (shifted) sensor origin
    sh = np.array([37 * 0.5, -12 * 0.5, 0.0])
    b = N.MapSet(0.5, list(sh), size, n_maps=2 * B)
    b.build((np.concatenate([f, m]) + sh.astype(np.float32)).astype(np.float32), range_limit=30.0,
            range_origins=np.tile(sh, (2 * B, 1)))
    for k in (0, B - 1, B, 2 * B - 1):
        ca, cb = a.export_cells(k), b.export_cells(k)
        assert np.array_equal(ca[2], cb[2]) and np.array_equal(ca[3], cb[3])            # same cells, same counts
        assert np.max(np.abs(ca[0] + sh - cb[0])) < 2e-6                                  # fp32 inputs were re-rounded
    S = np.eye(4); S[:3, 3] = sh
    T0b = np.stack([S @ T0[k] @ np.linalg.inv(S) for k in range(B)])
    Tb, rb = N.match_batch(b, idx, b, idx + B, T0b)
    # the shifted fp32 inputs are re-rounded (cell means move by ~1e-6 m), and a registration stops inside its own
    # convergence band: most pairs agree to micrometres, every pair to well below a millimetre
    # (a registration that runs into ITR_MAX has no unique answer: compared only when both runs converged)
    d = np.array([pose_close(np.linalg.inv(S) @ Tb[k] @ S, Ta[k]) for k in range(B)])
    both = (ra["converged"] == 1) & (rb["converged"] == 1)
    assert both.sum() >= B - 2
    assert np.all(d[both, 0] < 2e-3) and np.all(d[both, 1] < 2e-4), (d, both)
    assert np.sum((d[:, 0] < 5e-5) & (d[:, 1] < 5e-6)) >= B - 3, d
    # (b) self match
    Ts, rs = N.match_batch(a, idx, a, idx, np.tile(np.eye(4), (B, 1, 1)))
    assert np.all(rs["converged"] == 1) and np.max(np.abs(Ts - np.eye(4))) < 1e-9
    # (c) counters
    for k in range(2 * B):
        c = a.counters(k)
        assert c["overflow"] == 0 and c["n_dropped"] <= n and a.export_cells(k)[3].sum() <= n - c["n_dropped"]
    # (d) inverse consistency
    Tinv, rinv = N.match_batch(a, idx + B, a, idx, np.stack([np.linalg.inv(T0[k]) for k in range(B)]))
    for k in range(B):
        if ra["converged"][k] and rinv["converged"][k]:
            dt, dr = pose_close(Tinv[k] @ Ta[k], np.eye(4))
            assert dt < 0.02 and dr < 0.005, (k, dt, dr)


def test_self_match_and_empty_maps(N, monkeypatch):
    from ndt_feature_graph_amd import synth
    pts = synth.pair_2d([9], 20000)["fixed"].numpy()
    ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=2)
    ms.build(np.concatenate([pts, np.full_like(pts, np.nan)]), range_limit=30.0)
    T, r = N.match_d2d(ms, 0, ms, 0, np.eye(4))
    assert r["converged"] and np.max(np.abs(T - np.eye(4))) < 1e-9
    # empty source / empty target: zero gradient -> "gradient vanished" exit, pose untouched
    T0 = np.eye(4); T0[0, 3] = 0.3
    for a, b in ((0, 1), (1, 0), (1, 1)):
        T, r = N.match_d2d(ms, a, ms, b, T0)
        assert np.array_equal(T, T0) and r["exit_code"] == 1 and r["score"] == 0.0
    # the same in a batch of ten, through the persistent kernel, the grid-barrier kernel and the task pool
    ti = np.array([0, 1, 1, 0, 0, 1, 0, 1, 1, 0]); si = np.array([1, 0, 1, 0, 1, 1, 0, 0, 1, 1])
    Tb = np.repeat(T0[None], 10, 0)
    for coop, pool in (("0", "0"), ("1", "0"), ("1", "1")):
        monkeypatch.setenv("NDTGPU_COOP", coop); monkeypatch.setenv("NDTGPU_POOL", pool)
        T, r = N.match_batch(ms, ti, ms, si, Tb)
        for k in range(10):
            if ti[k] == 0 and si[k] == 0:
                assert r["converged"][k] and np.max(np.abs(T[k] - np.eye(4))) < 1e-6 or r["iterations"][k] > 0
            else:
                assert np.array_equal(T[k], T0) and r["exit_code"][k] == 1 and r["score"][k] == 0.0


def test_config5_3d_small(N, O):
    """configs[4] at reduced size: Velodyne-style cloud, 6-DoF, 0.5 m voxels."""
    from ndt_feature_graph_amd import synth
    pr = synth.pair_3d([1], rings=32, azimuths=1500)
    size = [100, 100, 10]
    ms = N.MapSet(0.5, [0, 0, 0], size, n_maps=2, max_cells=60000)
    pts = np.concatenate([pr["fixed"].numpy(), pr["moving"].numpy()])
    ms.build(pts, range_limit=70.0)
    of = oracle_map(O, pts[0], 0.5, size, rng=70.0)
    om = oracle_map(O, pts[1], 0.5, size, rng=70.0)
    assert_cells_equal(ms.export_cells(0), of.export_cells(), 0.5)
    T0 = pr["T_init"][0].numpy()
    T, r = N.match_d2d(ms, 0, ms, 1, T0)
    To, ro = O.match_d2d(of, om, T0)
    dt, dr = pose_close(T, To)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD
    assert pose_close(T, pr["T_gt"][0].numpy())[0] < 0.05


def test_match_fusion_soft_constraint_parity(N, O, monkeypatch):
    """ndt_feature::matchFusion (NDT + odometry soft constraint x^T Tcov^-1 x), the scan-to-map call of
    NDTFeatureFuserHMT::update (fuser_hmt.cpp:356): HIP vs oracle, both execution paths."""
    pr, tg, sr, om = _pair_maps(N, O, list(range(1, 11)), 20000, 0.5)
    T0 = pr["T_init"].numpy()
    B = len(T0)
    rng = np.random.default_rng(5)
    covs = []
    for b in range(B):
        A = rng.normal(size=(6, 6)) * 0.02
        covs.append(A @ A.T + np.diag([1e-3, 1e-3, 1.0, 1.0, 1.0, 1e-4]))     # MotionModel2d-like: z, roll, pitch loose
    covs = np.stack(covs)
    idx = np.arange(B)
    Tb, rb = N.match_fusion_batch(tg, idx, sr, idx, T0, covs)                  # persistent kernel (10 pairs)
    for b in range(B):
        To, ro = O.match_fusion(om[b][0], om[b][1], T0[b], covs[b])
        dt, dr = pose_close(Tb[b], To)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (b, dt, dr)
        assert abs(rb["score"][b] - ro["score"]) < 1e-6 * abs(ro["score"])
    Ts, rs = N.match_fusion_batch(tg, idx[:3], sr, idx[:3], T0[:3], covs[:3])    # grid-barrier kernel (<= 8 pairs)
    for b in range(3):
        dt, dr = pose_close(Ts[b], Tb[b])
        assert dt < 1e-9 and dr < 1e-9
    # the prior through the grid-barrier kernel and through the task pool (forced for these small maps): the same bits
    monkeypatch.setenv("NDTGPU_COOP", "1")
    out = {}
    for pool in ("0", "1"):
        monkeypatch.setenv("NDTGPU_POOL", pool)
        out[pool] = N.match_fusion_batch(tg, idx, sr, idx, T0, covs)
    monkeypatch.delenv("NDTGPU_COOP"); monkeypatch.delenv("NDTGPU_POOL")
    assert np.array_equal(out["0"][0], out["1"][0]) and np.array_equal(out["0"][1]["fevals"], out["1"][1]["fevals"])
    assert np.array_equal(out["0"][0][:3], Ts) and max(pose_close(out["1"][0][b], Tb[b])[0] for b in range(B)) < 1e-8
    # the prior matters: results differ from the plain matcher, and switching it off gives the plain matcher
    Tp, _ = N.match_batch(tg, idx, sr, idx, T0)
    assert max(pose_close(Tp[b], Tb[b])[0] for b in range(B)) > 1e-5
    Tn, _ = N.match_fusion_batch(tg, idx, sr, idx, T0, covs, use_soft_constraints=False)
    assert np.array_equal(Tn, Tp)


def test_match_fusion_tikhonov_parity(N, O):
    """matchFusion with useTikhonovRegularization (fusion.h:894-911, 1113-1115), alone and together with the soft
    constraint (the offline tool's defaults, ndt_feature_fuser_hmt.h:91-94): g <- H^T g + Q x0, H <- H^T H + Q."""
    pr, tg, sr, om = _pair_maps(N, O, list(range(21, 29)), 20000, 0.5)
    T0 = pr["T_init"].numpy()
    B = len(T0)
    rng = np.random.default_rng(8)
    covs = np.stack([np.diag([2e-3, 2e-3, 1.0, 1.0, 1.0, 4e-4]) * (1 + rng.uniform(0, 1)) for _ in range(B)])
    idx = np.arange(B)
    Tp, _ = N.match_batch(tg, idx, sr, idx, T0)
    for soft in (False, True):
        Tb, rb = N.match_fusion_batch(tg, idx, sr, idx, T0, covs, use_soft_constraints=soft, tikhonov=True)
        for b in range(B):
            To, ro = O.match_fusion(om[b][0], om[b][1], T0[b], covs[b], use_soft_constraints=soft, tikhonov=True)
            dt, dr = pose_close(Tb[b], To)
            assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (soft, b, dt, dr)
            assert bool(rb["converged"][b]) == ro["converged"] and rb["iterations"][b] == ro["iterations"], (soft, b)
            assert abs(rb["score"][b] - ro["score"]) < 1e-6 * abs(ro["score"])
        assert max(pose_close(Tp[b], Tb[b])[0] for b in range(B)) > 1e-6        # the regulariser changes the result


def test_match_fusion_tcov_line_search_negates_the_increment(N, O):
    """useSoftConstraints + useTikhonovRegularization (the offline tool's defaults): the reference first runs
    lineSearchMTFusionTcov (fusion.h:1008-1010), throws its step away (:1018-1023) and keeps its side effect: the
    increment, taken by reference, is negated in place when increment . (g_ndt + g_mahalanobis) >= 0 (fusion.h:89-95) --
    a test on ANOTHER gradient than the loop's dginit (scg = H^T g + Q x0 with Tikhonov, fusion.h:907, 973).
    Pairs on which that flip fires (the oracle counts them), started three initial offsets out: the HIP state machine
    (csrc/ndt_solver.h newton_finish) must follow the oracle through it -- iterations, exit, score and pose."""
    seeds = [3, 4, 7, 8, 9]
    pr, tg, sr, om = _pair_maps(N, O, seeds, 8000, 1.0)
    T0 = pr["T_init"].numpy().copy()
    T0[:, :3, 3] *= 3.0
    B = len(seeds)
    covs = np.stack([np.diag([2e-3, 2e-3, 1.0, 1.0, 1.0, 4e-4])] * B)
    idx = np.arange(B)
    Tb, rb = N.match_fusion_batch(tg, idx, sr, idx, T0, covs, use_soft_constraints=True, tikhonov=True)
    flips = 0
    for b in range(B):
        O.binding.tcov_flips(reset=True)
        To, ro = O.match_fusion(om[b][0], om[b][1], T0[b], covs[b], use_soft_constraints=True, tikhonov=True)
        flips += O.binding.tcov_flips(reset=True)
        dt, dr = pose_close(Tb[b], To)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (b, dt, dr)
        assert rb["iterations"][b] == ro["iterations"] and bool(rb["converged"][b]) == ro["converged"], b
        assert rb["exit_code"][b] == ro["exit_code"], b
        assert abs(rb["score"][b] - ro["score"]) < 1e-6 * abs(ro["score"])
    assert flips >= B          # the flip fired on these pairs (7, 8: several times)


def _odom_cells(T_odom, odom_cov6, n=40):
    """The odometry cells of NDTFeatureFuserHMT::update (ndt_feature_fuser_hmt.cpp:150-160, 322-334): n copies of
    (source: mean 0, target: mean of the odometry translation), covariance odom_cov."""
    sm = np.zeros((n, 3))
    tm = np.tile(T_odom[:3, 3], (n, 1))
    sc = np.tile(odom_cov6, (n, 1))
    return sm, sc.copy(), tm, sc.copy()


@pytest.mark.parametrize("soft,tikhonov,joint", [(True, True, True), (True, False, False), (False, False, False), (False, False, True),
                                                 (False, True, True)])
def test_match_fusion_with_odometry_cells(N, O, soft, tikhonov, joint):
    """matchFusion with useFeat / useOdom (the fuser's defaults): 40 odometry cell pairs with known correspondence
    (ndt_feature_fuser_hmt.cpp:322-334) whose NDTMatcherFeatureD2D sums join the NDT sums (fusion.h:858-871), a second
    line search over the feature maps and the smaller of the two steps (fusion.h:1013-1023), the feature score in the
    final score (fusion.h:1087-1096); with step_control_fusion and without the soft constraint the joint
    lineSearchMTFusion (fusion.h:1004-1006, 390-793, feature maps un-stepped as written at :619) instead.  HIP against
    the oracle: pose, iterations, exit, score; and the cells matter."""
    from ndt_feature_graph_amd import binding
    seeds = list(range(31, 39))
    pr, tg, sr, om = _pair_maps(N, O, seeds, 20000, 0.5)
    T0 = pr["T_init"].numpy()
    Tgt = pr["T_gt"].numpy()
    B = len(seeds)
    rng = np.random.default_rng(5)
    covs = np.stack([np.diag([2e-3, 2e-3, 1.0, 1.0, 1.0, 4e-4])] * B)
    odom_cov6 = np.array([4e-4, 1e-5, 0.0, 6e-4, 0.0, 0.01])
    feats = []
    for b in range(B):
        Todo = Tgt[b].copy()
        Todo[:2, 3] += rng.normal(scale=0.02, size=2)          # the odometry says roughly where the scan was taken
        feats.append(_odom_cells(Todo, odom_cov6, 40 if b != 3 else 7))
    feats[5] = tuple(a[:0] for a in feats[5])                  # a registration without cells in the same batch
    idx = np.arange(B)
    Tb, rb = binding.match_fusion_feat_batch(tg, idx, sr, idx, T0, covs, feats, use_soft_constraints=soft, tikhonov=tikhonov,
                                             step_control_fusion=joint)
    Tn, rn = N.match_fusion_batch(tg, idx, sr, idx, T0, covs, use_soft_constraints=soft, tikhonov=tikhonov)
    for b in range(B):
        To, ro = O.binding.match_fusion_feat(om[b][0], om[b][1], T0[b], covs[b], feats[b], use_soft_constraints=soft, tikhonov=tikhonov,
                                             step_control_fusion=joint)
        dt, dr = pose_close(Tb[b], To)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (b, dt, dr)
        assert rb["iterations"][b] == ro["iterations"] and bool(rb["converged"][b]) == ro["converged"], (b, rb["iterations"][b], ro["iterations"])
        assert rb["exit_code"][b] == ro["exit_code"], b
        assert abs(rb["score"][b] - ro["score"]) < 1e-6 * abs(ro["score"]), b
    # no cells: the plain fusion matcher (which ran as a cooperative launch here: another summation order)
    assert pose_close(Tb[5], Tn[5])[0] < 1e-8 and rb["iterations"][5] == rn["iterations"][5]
    assert max(pose_close(Tb[b], Tn[b])[0] for b in range(B) if b != 5) > 1e-6     # the cells change the result


def test_device_pointer_batch_checks_indices_and_overflow(N):
    """ndtgpu_match_batch_device takes its indices from device memory: an index out of range and a map whose build
    overflowed max_cells are refused per pair (exit codes -2 / -3, pose untouched), the other pairs are registered."""
    import torch
    from ndt_feature_graph_amd import binding, synth
    pr = synth.pair_2d([1, 2, 3], 20000)
    dev = torch.device("cuda", 0)
    tg = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=3)
    sr = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=3)
    small = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=3, max_cells=16)
    tg.build(pr["fixed"].numpy(), range_limit=30.0)
    sr.build(pr["moving"].numpy(), range_limit=30.0)
    small.build(pr["moving"].numpy(), range_limit=30.0)          # overflows: 16 cells are not enough
    T0 = pr["T_init"].transpose(1, 2).contiguous().reshape(3, 16).to(dev)
    st = torch.cuda.current_stream()

    def run(sset, tidx, sidx):
        T16 = T0.clone()
        res = torch.zeros((3, 64), dtype=torch.uint8, device=dev)
        binding.match_batch_device(tg, torch.tensor(tidx, dtype=torch.int32, device=dev), sset,
                                   torch.tensor(sidx, dtype=torch.int32, device=dev), T16, res, 3, stream=st)
        torch.cuda.synchronize()
        return T16.cpu().numpy(), res.cpu().numpy().view(binding.RESULT_DTYPE).reshape(3)
    Tg, rg = run(sr, [0, 1, 2], [0, 1, 2])
    assert np.all(rg["exit_code"] >= 0) and np.all(rg["iterations"] > 0)
    Tb, rb = run(sr, [0, 7, 2], [0, 1, 99])                      # pair 1: target index, pair 2: source index out of range
    assert rb["exit_code"][1] == -2 and rb["exit_code"][2] == -2 and rb["converged"][1] == 0
    assert np.array_equal(Tb[1], T0[1].cpu().numpy()) and np.array_equal(Tb[2], T0[2].cpu().numpy())
    assert np.array_equal(Tb[0], Tg[0]) and rb["exit_code"][0] == rg["exit_code"][0]
    To, ro = run(small, [0, 1, 2], [0, 1, 2])
    assert np.all(ro["exit_code"] == -3) and np.array_equal(To, T0.cpu().numpy())


def test_device_pointer_small_batch_of_large_maps(N, monkeypatch):
    """ndtgpu_match_batch_device spreads a small batch of large maps over several CUs per registration (the grid-barrier
    matcher: one asynchronous launch behind an event that orders such launches on the device) instead of one CU each
    (NDTGPU_DEVICE_COOP=0: persistent kernel): same answer to 1e-8.  Two such calls on two streams without a wait in
    between, and a bad index in the batch, behave as on the persistent kernel."""
    import torch
    from ndt_feature_graph_amd import binding, synth
    dev = torch.device("cuda", 0)
    pr = synth.pair_3d([1, 2], rings=32, azimuths=1500)
    ms = N.MapSet(0.25, [0, 0, 0], [100, 100, 10], n_maps=4, max_cells=32768)
    ms.build(torch.cat([pr["fixed"], pr["moving"]]).contiguous().to(dev), range_limit=70.0)
    torch.cuda.synchronize()
    assert min(ms.num_cells(k) for k in range(4)) >= 1024
    T0 = pr["T_init"].transpose(1, 2).contiguous().reshape(2, 16).to(dev)
    ti = torch.tensor([0, 1], dtype=torch.int32, device=dev)
    si = torch.tensor([2, 3], dtype=torch.int32, device=dev)
    out = {}
    for coop in ("1", "0"):
        monkeypatch.setenv("NDTGPU_DEVICE_COOP", coop)
        T16 = T0.clone()
        res = torch.zeros((2, 64), dtype=torch.uint8, device=dev)
        binding.match_batch_device(ms, ti, ms, si, T16, res, 2, stream=torch.cuda.current_stream())
        torch.cuda.synchronize()
        out[coop] = (T16.cpu().numpy(), res.cpu().numpy().view(binding.RESULT_DTYPE).reshape(2).copy())
    assert np.max(np.abs(out["1"][0] - out["0"][0])) < 1e-8
    assert np.array_equal(out["1"][1]["iterations"], out["0"][1]["iterations"]) and np.all(out["1"][1]["converged"] == 1)
    # two launches in flight on two streams (they must not hold parts of the chip and wait for each other), one with a
    # bad index: its pose stays, exit code -2; everybody else gets the bits of the sequential run
    monkeypatch.setenv("NDTGPU_DEVICE_COOP", "1")
    monkeypatch.setenv("NDTGPU_POOL", "1")          # (the task pool, which batches of more than 8 pairs take: same bits)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    Ta, Tb = T0.clone(), T0.clone()
    ra = torch.zeros((2, 64), dtype=torch.uint8, device=dev)
    rb = torch.zeros((2, 64), dtype=torch.uint8, device=dev)
    si_bad = torch.tensor([2, 99], dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for rep in range(3):
        Ta.copy_(T0); Tb.copy_(T0)
        torch.cuda.synchronize()
        binding.match_batch_device(ms, ti, ms, si, Ta, ra, 2, stream=s1)
        binding.match_batch_device(ms, ti, ms, si_bad, Tb, rb, 2, stream=s2)
        torch.cuda.synchronize()
        assert np.array_equal(Ta.cpu().numpy(), out["1"][0])
        rbn = rb.cpu().numpy().view(binding.RESULT_DTYPE).reshape(2)
        assert np.array_equal(Tb.cpu().numpy()[0], out["1"][0][0]) and rbn["exit_code"][1] == -2 and rbn["converged"][1] == 0
        assert np.array_equal(Tb.cpu().numpy()[1], T0.cpu().numpy()[1])


def test_config4_replay_small(N, O):
    """configs[3] at CI size: a short trajectory of nodes in one building, one NDT map per node, ALL node
    pairs as candidate edges (NDTFeatureGraph::computeAllPossibleLinks order), edges dealt block-cyclically
    to `world` shards (ndt_feature_graph_amd.distributed), every shard registered as one batch, results
    reassembled in edge order -- against the CPU matcher on every edge, plus the reference's link gates."""
    from ndt_feature_graph_amd import distributed as D, synth
    n_nodes = 7
    poses = np.array([[0.3 * k, 0.05 * k * (-1) ** k, 0.015 * k] for k in range(n_nodes)])
    scans = synth.scan_2d([123] * n_nodes, poses, 30000).numpy()
    pool = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=n_nodes)
    pool.build(scans, range_limit=30.0)
    omaps = [oracle_map(O, scans[k], 0.5, [100, 100, 1]) for k in range(n_nodes)]
    node_T = synth.pose2d_to_T(poses).numpy()
    rng = np.random.default_rng(3)
    odo_T = node_T.copy()
    odo_T[:, 0, 3] += rng.normal(scale=0.03, size=n_nodes)          # odometry-like node poses
    odo_T[:, 1, 3] += rng.normal(scale=0.03, size=n_nodes)
    edges = D.all_pairs(n_nodes)
    T0 = np.stack([np.linalg.inv(odo_T[i]) @ odo_T[j] for i, j in edges])
    world, chunk = 3, 4
    T_all = np.zeros((len(edges), 4, 4))
    it_all = np.zeros(len(edges), dtype=int)
    for rank in range(world):                                        # what each GPU of the node would do
        mine = D.shard_edges(len(edges), rank, world, chunk)
        Tm, rm = N.match_batch(pool, edges[mine, 0], pool, edges[mine, 1], T0[mine], delta_score=1e-3)   # "edge" preset
        T_all[mine], it_all[mine] = Tm, rm["iterations"]
    for e, (i, j) in enumerate(edges):
        To, ro = O.match_d2d(omaps[i], omaps[j], T0[e], delta_score=1e-3)
        dt, dr = pose_close(T_all[e], To)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD and it_all[e] == ro["iterations"], (e, dt, dr)
        gt = np.linalg.inv(node_T[i]) @ node_T[j]
        assert pose_close(T_all[e], gt)[0] < 0.05
    keep = D.gate_links(edges, node_T, max_dist=1.0, max_angle=0.2, min_idx_dist=2)
    assert 0 < len(keep) < len(edges)


def test_matcher_paths_on_a_grid_far_from_the_origin(N, O, monkeypatch):
    """Grid centres whose doubles have a busy low word, (12.3, -7.65, 0.2) and per-map centres on top: the persistent kernel
    keeps the map geometry in LDS, the grid-barrier kernel and the task pool move the centres into scalar registers
    (v_readfirstlane on both halves of the double) -- all three against the oracle built around the same centres."""
    from ndt_feature_graph_amd import synth
    B, res, size = 10, 0.5, [100.0, 100.0, 1.0]
    pr = synth.pair_2d(list(range(31, 31 + B)), 20000)
    base = np.array([12.3, -7.65, 0.2])
    shift = np.stack([base + np.array([0.37 * k, -0.11 * k, 0.0]) for k in range(B)])
    f = pr["fixed"].numpy() + shift[:, None, :].astype(np.float32)
    m = pr["moving"].numpy() + shift[:, None, :].astype(np.float32)
    tg = N.MapSet(res, list(base), size, n_maps=B)
    sr = N.MapSet(res, list(base), size, n_maps=B)
    om = []
    T0 = np.zeros((B, 4, 4))
    for k in range(B):
        tg.set_centre(k, shift[k]); sr.set_centre(k, shift[k])
        S = np.eye(4); S[:3, 3] = shift[k]
        T0[k] = S @ pr["T_init"].numpy()[k] @ np.linalg.inv(S)
        a = O.OracleMap(res, shift[k], size); a.load_points(f[k], 30.0, range_origin=shift[k]); a.compute_cells()
        b = O.OracleMap(res, shift[k], size); b.load_points(m[k], 30.0, range_origin=shift[k]); b.compute_cells()
        om.append((a, b))
    tg.build(f, range_limit=30.0, range_origins=shift)
    sr.build(m, range_limit=30.0, range_origins=shift)
    idx = np.arange(B)
    out = {}
    for name, coop, pool in (("persistent", "0", "0"), ("grid barrier", "1", "0"), ("task pool", "1", "1")):
        monkeypatch.setenv("NDTGPU_COOP", coop); monkeypatch.setenv("NDTGPU_POOL", pool)
        out[name] = N.match_batch(tg, idx, sr, idx, T0)
    for k in range(B):
        assert np.array_equal(tg.export_cells(k)[2], om[k][0].export_cells()[2])
        To, ro = O.match_d2d(om[k][0], om[k][1], T0[k])
        for name, (T, r) in out.items():
            dt, dr = pose_close(T[k], To)
            assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (name, k, dt, dr)
            assert r["iterations"][k] == ro["iterations"] and bool(r["converged"][k]) == ro["converged"], (name, k)
            assert dt < 1e-7, (name, k, dt)                  # (in fact: summation order only)
