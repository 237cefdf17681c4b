"""Randomised parity campaign of the node-map path (not part of the test suite): n node maps of `scans` scans each, fused on
the GPU (ndtgpu_mapset_add_cloud, all nodes per launch) and in the oracle (addPointCloud + computeNDTCells per scan, in the
reference's beam-after-beam order AND in the order-free form the GPU implements).
usage (GPU box): python tools/parity_campaign_fuse.py [nodes=200] [scans=10] [points=20000]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import synth
from oracle import binding as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_scans = int(sys.argv[2]) if len(sys.argv) > 2 else 10
npts = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
res, size = 0.5, [100.0, 100.0, 1.0]


def node_scans(seed):
    poses = np.array([[0.15 * k, 0.05 * np.sin(k + seed), 0.02 * k] for k in range(n_scans)])
    scans = synth.scan_2d([seed] * n_scans, poses, npts).numpy()
    T = synth.pose2d_to_T(poses).numpy()
    T = np.linalg.inv(T[0]) @ T
    out = np.empty_like(scans)
    for k in range(n_scans):
        out[k] = (scans[k].astype(np.float64) @ T[k][:3, :3].T + T[k][:3, 3]).astype(np.float32)
    return out, T[:, :3, 3].copy()


seeds = list(range(9000, 9000 + n))
clouds = np.empty((n, n_scans, npts, 3), np.float32)
origins = np.empty((n, n_scans, 3))
for i, s in enumerate(seeds):
    clouds[i], origins[i] = node_scans(s)
ms = N.MapSet(res, [0, 0, 0], size, n_maps=n, max_cells=4096)
ms.enable_occupancy()
for k in range(n_scans):
    kw = dict(maxz=100.0, sensor_noise=0.1) if k == 0 else dict(maxz=25.0, sensor_noise=0.06)
    ms.add_cloud(np.ascontiguousarray(clouds[:, k]), np.ascontiguousarray(origins[:, k]), **kw)
t0 = time.time()
bad = {"free": 0, "seq": 0}
worst = {"free": [0.0, 0.0, 0.0], "seq": [0.0, 0.0, 0.0]}
for i in range(n):
    g = ms.export_cells(i)
    og = ms.occupancy(i)
    for name, order_free in (("free", True), ("seq", False)):
        om = O.OracleMap(res, [0, 0, 0], size)
        for k in range(n_scans):
            kw = dict(maxz=100.0, sensor_noise=0.1) if k == 0 else dict(maxz=25.0, sensor_noise=0.06)
            om.add_point_cloud(origins[i, k], clouds[i, k], order_free=order_free, **kw)
            om.compute_cells_full()
        o = om.export_cells()
        if len(g[3]) != len(o[3]) or not np.array_equal(g[2], o[2]) or not np.array_equal(g[3].astype(np.int64), o[3].astype(np.int64)):
            bad[name] += 1
            continue
        w = worst[name]
        if len(g[3]):
            w[0] = max(w[0], float(np.abs(g[0] - o[0]).max()))
            w[1] = max(w[1], float((np.abs(g[1] - o[1]) / np.abs(o[1]).max(axis=(1, 2), keepdims=True)).max()))
        w[2] = max(w[2], float(np.abs(og - om.occupancy()).max()))
print("%d node maps x %d scans x %d points, %.0f cells per map (%.0f s of oracle)" % (n, n_scans, npts, ms.num_cells_all().mean(), time.time() - t0))
for name, what in (("free", "oracle, order-free form"), ("seq", "oracle, the reference's beam-after-beam order")):
    w = worst[name]
    print("  against the %s: node maps with another cell set or another N in a cell: %d; worst |mean| %.2e m, covariance %.2e relative, occupancy %.2e" % (
        what, bad[name], w[0], w[1], w[2]))
