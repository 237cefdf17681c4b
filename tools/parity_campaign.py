"""Randomised parity campaign (not part of the test suite): n 2D scan pairs, HIP path against the oracle.
usage (GPU box): python tools/parity_campaign.py 1500"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import synth
from oracle import binding as O
n, npts, res, size, rng = int(sys.argv[1]), 20000, 0.5, [100.0, 100.0, 1.0], 30.0
seeds = list(range(5000, 5000 + n))
pr = synth.pair_2d(seeds, npts)
fixed, moving = pr["fixed"].numpy(), pr["moving"].numpy()
T0 = pr["T_init"].numpy()
ms = N.MapSet(res, [0, 0, 0], size, n_maps=2 * n, max_cells=4096)
ms.build(torch.from_numpy(np.concatenate([fixed, moving])).cuda(), range_limit=rng); torch.cuda.synchronize()
idx = np.arange(n)
T, r = N.match_batch(ms, idx, ms, idx + n, T0)
worst = [0.0, 0.0]; it_diff = 0; conv_diff = 0; big = []
t0 = time.time()
for k in range(n):
    a = O.OracleMap(res, [0, 0, 0], size); a.load_points(fixed[k], rng); a.compute_cells()
    b = O.OracleMap(res, [0, 0, 0], size); b.load_points(moving[k], rng); b.compute_cells()
    To, ro = O.match_d2d(a, b, T0[k])
    dt = float(np.linalg.norm(T[k][:3, 3] - To[:3, 3])); dr = float(np.linalg.norm(T[k][:3, :3] - To[:3, :3]))
    worst = [max(worst[0], dt), max(worst[1], dr)]
    it_diff += int(r["iterations"][k] != ro["iterations"]); conv_diff += int(bool(r["converged"][k]) != ro["converged"])
    if dt > 1e-6 or dr > 1e-6: big.append((seeds[k], dt, dr, int(r["iterations"][k]), ro["iterations"]))
print("%d pairs: worst |dt| %.3e m, worst |dR| %.3e; iteration counts differ on %d, convergence flags on %d; pairs beyond 1e-6: %s (%.0f s of oracle)" % (
    n, worst[0], worst[1], it_diff, conv_diff, big[:8], time.time() - t0))
