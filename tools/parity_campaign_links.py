"""Parity campaign of the per-link outputs of NDTFeatureGraph::updateLinksUsingNDTRegistration (not part of the test suite):
EVERY gated candidate edge of a 500-node replay on fused node maps (the layout of tests/test_gpu_replay.py: 5 rooms x 100
nodes, 3 scans of 6000 points per node) -- registration (edge preset), NDTMatcherD2D::covariance and
overlapNDTOccupancyScore -- HIP path against the oracle on oracle-built node maps.
usage (GPU box): python tools/parity_campaign_links.py [nodes=500]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import distributed as D, synth
from oracle import binding as O
from test_gpu_replay import replay_layout

dev = torch.device("cuda", 0)
n_nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 500
S, n_pts, res, size = 3, 6000, 0.5, [100.0, 100.0, 1.0]
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
    sc[:, :, 0] += dx
    clouds.append((sc, np.tile(np.array([[dx, 0.0, 0.0]]), (n_nodes, 1))))
fuse_kw = [dict(maxz=100.0, sensor_noise=0.1)] + [dict(maxz=25.0, sensor_noise=0.06)] * (S - 1)
ms = N.MapSet(res, [0, 0, 0], size, n_maps=n_nodes, max_cells=2048)
ms.enable_occupancy()
for (sc, org), kw in zip(clouds, fuse_kw):
    ms.add_cloud(sc, org, **kw)
edges = D.all_pairs(n_nodes)
d_odo = np.linalg.norm(odo_T[edges[:, 0], :2, 3] - odo_T[edges[:, 1], :2, 3], axis=1)
gi = np.nonzero((d_odo <= 6.0) & ((edges[:, 1] - edges[:, 0]) >= 2))[0]
T0 = np.einsum("eij,ejk->eik", np.linalg.inv(odo_T)[edges[gi, 0]], odo_T[edges[gi, 1]])
Tg, rg = N.match_batch(ms, edges[gi, 0], ms, edges[gi, 1], T0, delta_score=1e-3)
cov, sing = N.covariance(ms, edges[gi, 0], ms, edges[gi, 1], Tg)
score, nb = N.overlap_score(ms, edges[gi, 0], ms, edges[gi, 1], Tg)
t0 = time.time()
scans_h = [(sc.cpu().numpy(), org) for sc, org in clouds]
omaps, cells_bad = [], 0
for k in range(n_nodes):
    om = O.OracleMap(res, [0, 0, 0], size)
    for (sc, org), kw in zip(scans_h, fuse_kw):
        om.add_point_cloud(org[k], sc[k], maxz=kw["maxz"], sensor_noise=kw["sensor_noise"], order_free=True)
        om.compute_cells_full()
    a, b = ms.export_cells(k), om.export_cells()
    cells_bad += int(len(a[3]) != len(b[3]) or not np.array_equal(a[2], b[2]) or not np.array_equal(a[3].astype(np.int64), b[3].astype(np.int64)))
    omaps.append(om)
worst = [0.0, 0.0, 0.0, 0.0]; itd = cvd = nbd = beyond = n_cov = 0; loose = []; worst_conv = 0.0
for q, e in enumerate(gi):
    i, j = (int(v) for v in edges[e])
    To, ro = O.match_d2d(omaps[i], omaps[j], T0[q], delta_score=1e-3)
    dt = float(np.linalg.norm(Tg[q][:3, 3] - To[:3, 3])); dr = float(np.linalg.norm(Tg[q][:3, :3] - To[:3, :3]))
    worst[0], worst[1] = max(worst[0], dt), max(worst[1], dr)
    itd += int(rg["iterations"][q] != ro["iterations"]); cvd += int(bool(rg["converged"][q]) != bool(ro["converged"])); beyond += int(dt > 1e-6 or dr > 1e-6)
    if rg["converged"][q]: worst_conv = max(worst_conv, dt)
    if dt > 1e-6 or dr > 1e-6:
        # the scatter of the ORACLE's own poses on this edge under another summation order / rounding-level noise
        best, spread, flow = dt, 0.0, False
        for mode in range(1, 28):
            O.set_sum_mode(mode)
            Tm, rm = O.match_d2d(omaps[i], omaps[j], T0[q], delta_score=1e-3)
            best = min(best, float(np.linalg.norm(Tg[q][:3, 3] - Tm[:3, 3]))); spread = max(spread, float(np.linalg.norm(To[:3, 3] - Tm[:3, 3])))
            flow = flow or rm["iterations"] != ro["iterations"]
        O.set_sum_mode(0)
        loose.append((i, j, "%.1e" % dt, "nearest %.1e" % best, "scatter %.1e" % spread, flow, int(rg["iterations"][q]), ro["iterations"], int(rg["converged"][q])))
    if not sing[q]:
        co = O.covariance(omaps[i], omaps[j], Tg[q])
        worst[2] = max(worst[2], float(np.abs(cov[q] - co).max() / np.abs(co).max())); n_cov += 1
    so, nbo = O.overlap_score(omaps[i], omaps[j], Tg[q])
    nbd += int(nb[q] != nbo); worst[3] = max(worst[3], abs(score[q] - so) / max(so, 1e-3))
print("%d fused node maps (%d with another cell set than the oracle's), %d gated edges: worst |dt| %.2e m |dR| %.2e; iteration counts differ on %d, convergence flags on %d, poses beyond 1e-6 on %d; covariance (%d links with a regular Hessian) worst %.1e relative; overlap: neighbour counts differ on %d, score worst %.1e relative (%.0f s of oracle)" % (
    n_nodes, cells_bad, len(gi), worst[0], worst[1], itd, cvd, beyond, n_cov, worst[2], nbd, worst[3], time.time() - t0))
print("converged edges: worst |dt| %.2e m; edges beyond 1e-6 (i, j, |dt|, nearest of the oracle's poses under its 27 modes, their scatter, oracle flow changes, iterations hip / oracle, converged):" % worst_conv, loose)
