#!/usr/bin/env python3
"""Timeline of one launch of the persistent matcher (library built with -DNDT_MATCH_PROF; NDTGPU_LIB names it).
Prints when the tickets run out, when workgroups leave, how long the ITR_MAX registrations take alone on a CU, and the
section clocks of the share tasks.  usage (GPU box): NDTGPU_LIB=.../libndtgpu_prof.so python tools/timeline_match.py"""
import argparse, ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import binding, synth

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=1024)
ap.add_argument("--points", type=int, default=100000)
a = ap.parse_args()
dev = torch.device("cuda", 0)
B = a.pairs
pr = synth.pair_2d(torch.arange(1, B + 1, device=dev), a.points, device=dev, chunk_bytes=2 << 30)
Ti = pr["T_init"].transpose(1, 2).contiguous().reshape(B, 16)
ts = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=B, max_cells=4096)
ss = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=B, max_cells=4096)
st = torch.cuda.current_stream()
ts.build(pr["fixed"].contiguous(), range_limit=30.0, stream=st); ss.build(pr["moving"].contiguous(), range_limit=30.0, stream=st)
ts.profiling(True)
L = binding.lib()
NTL = 4 * 4096 + 1024 + 8
tl = np.zeros(NTL, dtype=np.int64)
prof = np.zeros(16, dtype=np.int64)
HAVE_PROF = hasattr(L, 'ndtgpu_debug_prof')
sprof = np.zeros(16, dtype=np.int64); sp = sprof.ctypes.data_as(C.POINTER(C.c_longlong))
tlp = tl.ctypes.data_as(C.POINTER(C.c_longlong)); pp = prof.ctypes.data_as(C.POINTER(C.c_longlong))


def run(idx_t, tag):
    n = len(idx_t)
    idx = torch.tensor(idx_t, dtype=torch.int32, device=dev)
    T16 = Ti[idx.long()].clone()
    res = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
    for rep in range(2):                                     # second run is the measured one
        T16.copy_(Ti[idx.long()])
        L.ndtgpu_debug_timeline(tlp, 1)
        if HAVE_PROF: L.ndtgpu_debug_prof(pp, 1); L.ndtgpu_debug_solver_prof(sp, 1)
        binding.match_batch_device(ts, idx, ss, idx, T16, res, n, stream=st)
        torch.cuda.synchronize()
    ms = ts.last_kernel_ms(1)
    L.ndtgpu_debug_timeline(tlp, 0)
    if HAVE_PROF:
        L.ndtgpu_debug_prof(pp, 0); L.ndtgpu_debug_solver_prof(sp, 0)
        names = ["assemble", "factor", "regularize", "ldlt", "finish", "apply_step", "request_trial", "linesearch"]
        print("   solver stages (cycles per call x calls): " + "  ".join("%s %.0f x %d" % (names[k], sprof[2 * k] / max(1, sprof[2 * k + 1]), sprof[2 * k + 1]) for k in range(8)))
    r = res.cpu().numpy().view(binding.RESULT_DTYPE).reshape(n)
    t0 = tl[4 * 4096 + 1024]
    us = lambda x: (x - t0) / 100.0
    P = tl[:4 * 4096].reshape(4096, 4)[:n]
    start, park, resume, end = us(P[:, 0]), P[:, 1], P[:, 2], us(P[:, 3])
    parked = park > 0
    wg = tl[4 * 4096:4 * 4096 + 1024]; wg = us(wg[wg > 0])
    print("== %s: %d pairs, kernel %.3f ms" % (tag, n, ms))
    print("   last fresh start %.0f us; pairs parked %d; first resume %.0f us, last resume %.0f us" % (
        start.max(), parked.sum(), us(resume[parked]).min() if parked.any() else -1, us(resume[parked]).max() if parked.any() else -1))
    print("   pair end percentiles 50/90/99/100: %s us" % np.percentile(end, [50, 90, 99, 100]).round(0))
    print("   workgroups %d; exit time percentiles 10/50/90/100: %s us" % (len(wg), np.percentile(wg, [10, 50, 90, 100]).round(0)))
    print("   workgroups still resident at 25/50/75 %% of the launch: %s" % [int((wg > f * wg.max()).sum()) for f in (0.25, 0.5, 0.75)])
    long = r["iterations"] >= 31
    if long.any():
        dur_res = end[long] - np.where(parked[long], us(resume[long]), start[long])
        print("   ITR_MAX pairs %d: resumed at %.0f..%.0f us, time from (re)start to end mean %.0f max %.0f us, fevals mean %.0f" % (
            long.sum(), us(resume[long & parked]).min() if (long & parked).any() else -1, us(resume[long & parked]).max() if (long & parked).any() else -1,
            dur_res.mean(), dur_res.max(), r["fevals"][long].mean()))
    for base, name in ((0, "gradient-only"), (8, "with Hessian")):
        c = prof[base + 5]
        if c:
            v = prof[base:base + 5] / c
            print("   share task %-13s (lane 0 clocks, %d tasks): transform %.0f  probe %.0f  list %.0f  term %.0f  rest %.0f  = %.0f" % (
                name, c, v[0], v[1], v[2], v[3], v[4], v.sum()))
    print("   solver: wave cycles per iteration %.0f; share wave-cycles/8 per eval %.0f; eval wall (publish -> last share) per eval %.0f" % (
        r["cycles_solver"].sum() / max(1, r["iterations"].sum()), r["cycles_eval"].sum() / max(1, r["fevals"].sum()),
        r["pair_terms_g"].sum() / max(1, r["fevals"].sum())))
    k = int(np.argmax(r["fevals"]))
    tot_us = end[k] - (us(resume[k]) if parked[k] else start[k])
    print("   longest pair %d: fevals %d iters %d: eval wall %.0f k + solver %.0f k cycles; (re)start -> end %.0f us  => %.2f GHz if nothing else" % (
        k, r["fevals"][k], r["iterations"][k], r["pair_terms_g"][k] / 1e3, r["cycles_solver"][k] / 1e3, tot_us,
        (r["pair_terms_g"][k] + r["cycles_solver"][k]) / max(tot_us, 1e-9) / 1e3))
    return r


r = run(list(range(B)), "full batch")
long = np.nonzero(r["iterations"] >= 31)[0]
if len(long):
    run(list(long), "ITR_MAX pairs only (one per CU: the tail alone)")
    run(list(long[:1]), "one ITR_MAX pair")
med = np.nonzero((r["iterations"] >= 6) & (r["iterations"] <= 7))[0][:256]
run(list(med), "256 median pairs, one per CU")
