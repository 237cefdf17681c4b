"""Randomised parity campaign (not part of the test suite): n 2D scan pairs, HIP path against the oracle.
usage (GPU box): python tools/parity_campaign.py 1500 [6dof|3dof] [points=20000]   (3dof: NDTMatcherD2D_2D, dof_mask 0x23)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import synth
from oracle import binding as O
n, npts, res, size, rng = int(sys.argv[1]), (int(sys.argv[3]) if len(sys.argv) > 3 else 20000), 0.5, [100.0, 100.0, 1.0], 30.0
kw = dict(dof_mask=0x23) if len(sys.argv) > 2 and sys.argv[2] == "3dof" else {}
seeds = list(range(5000, 5000 + n))
pr = synth.pair_2d(seeds, npts)
fixed, moving = pr["fixed"].numpy(), pr["moving"].numpy()
T0 = pr["T_init"].numpy()
ms = N.MapSet(res, [0, 0, 0], size, n_maps=2 * n, max_cells=4096)
ms.build(torch.from_numpy(np.concatenate([fixed, moving])).cuda(), range_limit=rng); torch.cuda.synchronize()
idx = np.arange(n)
T, r = N.match_batch(ms, idx, ms, idx + n, T0, **kw)
worst = [0.0, 0.0]; it_diff = 0; conv_diff = 0; big = []; flow_diff = 0; flow_chaotic = 0; flow_bad = []; hip_only = []; cells_bad = 0; cell_worst = [0.0, 0.0]
t0 = time.time()
for k in range(n):
    a = O.OracleMap(res, [0, 0, 0], size); a.load_points(fixed[k], rng); a.compute_cells()
    b = O.OracleMap(res, [0, 0, 0], size); b.load_points(moving[k], rng); b.compute_cells()
    for mi, om in ((k, a), (n + k, b)):                         # the two maps, cell by cell
        gc, oc = ms.export_cells(mi), om.export_cells()
        if len(gc[3]) != len(oc[3]) or not np.array_equal(gc[2], oc[2]) or not np.array_equal(gc[3].astype(np.int64), oc[3].astype(np.int64)):
            cells_bad += 1
        elif len(gc[3]):
            cell_worst[0] = max(cell_worst[0], float(np.abs(gc[0] - oc[0]).max()))
            cell_worst[1] = max(cell_worst[1], float((np.abs(gc[1] - oc[1]) / np.abs(oc[1]).max(axis=(1, 2), keepdims=True)).max()))
    To, ro = O.match_d2d(a, b, T0[k], **kw)
    dt = float(np.linalg.norm(T[k][:3, 3] - To[:3, 3])); dr = float(np.linalg.norm(T[k][:3, :3] - To[:3, :3]))
    worst = [max(worst[0], dt), max(worst[1], dr)]
    it_diff += int(r["iterations"][k] != ro["iterations"]); conv_diff += int(bool(r["converged"][k]) != ro["converged"])
    if dt > 1e-6 or dr > 1e-6: big.append((seeds[k], dt, dr, int(r["iterations"][k]), ro["iterations"]))
    if r["iterations"][k] != ro["iterations"] or r["exit_code"][k] != ro["exit_code"] or bool(r["converged"][k]) != ro["converged"]:
        # a control-flow difference counts as chaos only where the ORACLE ALONE changes its flow under another summation order /
        # ulp noise on its sums / cond(H) eps on its Newton increments
        flow_diff += 1
        base, chaotic = (ro["iterations"], ro["exit_code"], ro["converged"]), False
        for mode in list(range(16, 28)) + list(range(1, 16)):
            O.set_sum_mode(mode)
            Tq, rq = O.match_d2d(a, b, T0[k], **kw)
            if (rq["iterations"], rq["exit_code"], rq["converged"]) != base:
                chaotic = True
                break
        O.set_sum_mode(0)
        if not chaotic:
            # ... or where the HIP path alone does: the same pair through the grid-barrier kernel (another summation order: chunks
            # instead of shares; calls of <= 8 pairs take it)
            Tq, rq = N.match_batch(ms, [k], ms, [k + n], T0[k:k + 1], **kw)
            if (rq["iterations"][0], rq["exit_code"][0]) != (r["iterations"][k], r["exit_code"][k]):
                chaotic = True
                hip_only.append(seeds[k])
        flow_chaotic += chaotic
        if not chaotic: flow_bad.append(seeds[k])
print("%d maps: %d with another cell set or another point count in a cell than the oracle's; means within %.1e m, covariances within %.1e relative" % (2 * n, cells_bad, cell_worst[0], cell_worst[1]))
print("%d pairs%s: worst |dt| %.3e m, worst |dR| %.3e; iteration counts differ on %d, convergence flags on %d; pairs beyond 1e-6: %d %s; control flow differs on %d, on %d of them the oracle alone changes its flow under noise or (%s) the HIP path alone between its two summation orders, not on: %s; converged %.2f (%.0f s of oracle)" % (
    n, " (3-DoF)" if kw else "", worst[0], worst[1], it_diff, conv_diff, len(big), big[:6], flow_diff, flow_chaotic, hip_only, flow_bad, r["converged"].mean(), time.time() - t0))
