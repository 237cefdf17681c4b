"""Randomised 3D parity campaign (not part of the test suite): n 3D sweep pairs, HIP path (build + match) against the oracle.
usage (GPU box): python tools/parity_campaign_3d.py 40 [rings=32] [azimuths=1500]   (64 3125: the full-size 200 k-point sweeps)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import synth
from oracle import binding as O
n = int(sys.argv[1]); seeds = list(range(300, 300 + n))
RINGS = int(sys.argv[2]) if len(sys.argv) > 2 else 32
AZ = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
pr = synth.pair_3d(seeds, rings=RINGS, azimuths=AZ)
fixed, moving, T0 = pr["fixed"].numpy(), pr["moving"].numpy(), pr["T_init"].numpy()
res, size, rng = 0.25, [100.0, 100.0, 10.0], 70.0
ms = N.MapSet(res, [0, 0, 0], size, n_maps=2 * n, max_cells=32768)
ms.build(np.concatenate([fixed, moving]), range_limit=rng)
idx = np.arange(n)
T, r = N.match_batch(ms, idx, ms, idx + n, T0)
worst = [0, 0]; itd = 0; cells_bad = 0; big = []; worst_conv = 0.0
for k in range(n):
    a = O.OracleMap(res, [0, 0, 0], size); a.load_points(fixed[k], rng); a.compute_cells()
    b = O.OracleMap(res, [0, 0, 0], size); b.load_points(moving[k], rng); b.compute_cells()
    ga, oa = ms.export_cells(k), a.export_cells()
    if len(ga[3]) != len(oa[3]) or not np.array_equal(ga[2], oa[2]) or not np.array_equal(ga[3], oa[3]): cells_bad += 1
    To, ro = O.match_d2d(a, b, T0[k])
    worst = [max(worst[0], float(np.linalg.norm(T[k][:3, 3] - To[:3, 3]))), max(worst[1], float(np.linalg.norm(T[k][:3, :3] - To[:3, :3])))]
    itd += int(r["iterations"][k] != ro["iterations"])
    dt_k = float(np.linalg.norm(T[k][:3, 3] - To[:3, 3]))
    if r["converged"][k]: worst_conv = max(worst_conv, dt_k)
    if dt_k > 1e-6: big.append((seeds[k], "%.2e" % dt_k, int(r["iterations"][k]), ro["iterations"], int(r["converged"][k]), int(ro["converged"]), int(r["exit_code"][k]) if "exit_code" in r.dtype.names else -9, float(r["score"][k]) if "score" in r.dtype.names else 0.0, ro.get("score", 0.0)))
print("%d 3D pairs (%d points, 0.25 m): maps with a different cell set %d; worst |dt| %.2e m |dR| %.2e (among the converged: %.2e m); iteration counts differ on %d; converged %.2f" % (n, RINGS * AZ, cells_bad, worst[0], worst[1], worst_conv, itd, r["converged"].mean()))
print("pairs beyond 1e-6 m (seed, |dt|, iterations hip / oracle, converged hip / oracle, exit code, score hip / oracle):", big)
