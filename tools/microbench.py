#!/usr/bin/env python3
"""Kernel-level timings on one GPU (HIP events inside the library).  Usage on the GPU box:
   python tools/microbench.py [--pairs 1024] [--points 100000]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import binding, synth

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=1024)
ap.add_argument("--points", type=int, default=100000)
ap.add_argument("--res", type=float, default=0.5)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda", 0)
B = a.pairs
pr = synth.pair_2d(torch.arange(1, B + 1, device=dev), a.points, device=dev, chunk_bytes=2 << 30)
fixed, moving = pr["fixed"].contiguous(), pr["moving"].contiguous()
Ti = pr["T_init"].transpose(1, 2).contiguous().reshape(B, 16)
T16 = Ti.clone()
res = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
idx = torch.arange(B, dtype=torch.int32, device=dev)
ts = N.MapSet(a.res, [0, 0, 0], [100, 100, 1], n_maps=B, max_cells=4096)
ss = N.MapSet(a.res, [0, 0, 0], [100, 100, 1], n_maps=B, max_cells=4096)
ts.profiling(True); ss.profiling(True)
st = torch.cuda.current_stream()

def t_build():
    v = []
    for _ in range(a.reps):
        ts.build(fixed, range_limit=30.0, stream=st); ss.build(moving, range_limit=30.0, stream=st)
        v.append((ts.last_kernel_ms(0) + ss.last_kernel_ms(0)) / 2)
    return min(v)

def t_match(**kw):
    v = []
    for _ in range(a.reps):
        T16.copy_(Ti)
        binding.match_batch_device(ts, idx, ss, idx, T16, res, B, stream=st, **kw)
        v.append(ts.last_kernel_ms(1))
    r = res.cpu().numpy().view(binding.RESULT_DTYPE).reshape(B)
    return min(v), r

bm = t_build()
gb = (12.0 * a.points * B) / bm / 1e6
cy = np.array([ts.counters(i)["cyc"] for i in range(0, B, max(1, B // 32))])
print("build phases (k cycles, mean over sampled maps): A %.0f  B %.0f  C %.0f  D %.0f" % tuple(cy.mean(axis=0) / 1e3))
print("build: %.3f ms per %d scans  (%.1f GB/s of point bytes, %.2f us/scan)" % (bm, B, gb, 1e3 * bm / B))
for name, kw in [("full", {}), ("no step control", dict(step_control=0)), ("itr_max=0 (3 newton iters)", dict(itr_max=0)),
                 ("nn=1", dict(n_neighbours=1)), ("nn=0", dict(n_neighbours=0)), ("3dof", dict(dof_mask=0x23))]:
    ms, r = t_match(**kw)
    fe = r["fevals"].sum(); it = r["iterations"].sum()
    print("match %-28s %.3f ms  fevals %d iters %d  -> %.2f us per (pair-eval)  M_src %.0f" % (
        name, ms, fe, it, 1e3 * ms / max(fe, 1) * 1.0, r["n_source"].mean()))
ms, r = t_match()
clk = 2.4e9
print("cycles: eval mean %.0f k  solver mean %.0f k per pair; per eval %.1f k, per iter solver %.1f k" % (
    r["cycles_eval"].mean() / 1e3, r["cycles_solver"].mean() / 1e3, r["cycles_eval"].sum() / r["fevals"].sum() / 1e3,
    r["cycles_solver"].sum() / max(r["iterations"].sum(), 1) / 1e3))
print("pair total cycles: mean %.0f k max %.0f k; sum/256CU = %.3f ms at 2.4GHz" % (
    (r["cycles_eval"] + r["cycles_solver"]).mean() / 1e3, (r["cycles_eval"] + r["cycles_solver"]).max() / 1e3,
    (r["cycles_eval"] + r["cycles_solver"]).sum() / 256 / clk * 1e3))
tot = (r["cycles_eval"] + r["cycles_solver"]).astype(np.float64)
order = np.argsort(-tot)
print("slowest pairs: (idx, kcycles, iterations, fevals, pair_terms_g+h, converged)")
for i in order[:8]:
    print("   ", i, int(tot[i] / 1e3), r["iterations"][i], r["fevals"][i], r["pair_terms_g"][i] + r["pair_terms_h"][i], r["converged"][i])
q = np.percentile(tot, [50, 90, 95, 99, 100]) / 1e3
print("pair kcycles percentiles 50/90/95/99/100:", q.round(0), " fevals pct:", np.percentile(r["fevals"], [50, 90, 95, 99, 100]))
print("iterations histogram:", np.bincount(r["iterations"], minlength=32))
