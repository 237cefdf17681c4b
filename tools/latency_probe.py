#!/usr/bin/env python3
"""Where a single-pair call (the reference's call shape: build x2 + match, host-synchronous) spends its time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import binding, synth
dev = torch.device("cuda", 0)
pr = synth.pair_2d(torch.tensor([1], device=dev), 100000, device=dev)
scans = torch.stack([pr["fixed"][0], pr["moving"][0]]).contiguous()
T0 = pr["T_init"][0].cpu().numpy()
ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=2, max_cells=4096)
ms.profiling(True)
st = torch.cuda.current_stream()
keep = []
if os.environ.get("LAT_STREAMS"):                      # other streams with finished work in the process, like bench.py's pipeline
    for _ in range(int(os.environ["LAT_STREAMS"])):
        q = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(q):
            keep.append(torch.zeros(1024, device=dev) + 1)
        keep.append(q)
    torch.cuda.synchronize()
if os.environ.get("LAT_MAPSETS"):                      # a large map set that was built once
    big = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=int(os.environ["LAT_MAPSETS"]), max_cells=4096)
    big.build(scans.repeat(int(os.environ["LAT_MAPSETS"]) // 2, 1, 1).contiguous(), range_limit=30.0, stream=st)
    torch.cuda.synchronize()
    keep.append(big)
if os.environ.get("LAT_SLEEP"):
    time.sleep(float(os.environ["LAT_SLEEP"]))        # an idle GPU first (clocks down), like bench.py's latency leg after the CPU baseline
tb, tm, tt = [], [], []
for k in range(30):
    torch.cuda.synchronize()
    c0 = time.perf_counter()
    ms.build(scans, range_limit=30.0, stream=st)
    c1 = time.perf_counter()
    torch.cuda.synchronize()
    c2 = time.perf_counter()
    T, r = N.match_d2d(ms, 0, ms, 1, T0)
    c3 = time.perf_counter()
    tb.append((c1 - c0, c2 - c0)); tm.append(c3 - c2); tt.append(c3 - c0)
tb = np.array(tb[5:]) * 1e3; tm = np.array(tm[5:]) * 1e3; tt = np.array(tt[5:]) * 1e3
print("single 2D pair, 100 k points: build call returns after %.3f ms, build done after %.3f ms (kernel events %.3f ms); match_d2d %.3f ms; total %.3f ms (median of 25)" % (
    np.median(tb[:, 0]), np.median(tb[:, 1]), ms.last_kernel_ms(0), np.median(tm), np.median(tt)))
print("iterations", int(r["iterations"]), "fevals", int(r["fevals"]), "cycles eval/solver (workgroup 0)", int(r["cycles_eval"]), int(r["cycles_solver"]))
# the same without the sync between build and match (what bench.py's latency leg does)
tt2 = []
for k in range(30):
    torch.cuda.synchronize()
    c0 = time.perf_counter()
    ms.build(scans, range_limit=30.0, stream=st)
    T, r = N.match_d2d(ms, 0, ms, 1, T0)
    tt2.append(time.perf_counter() - c0)
print("back to back: %.3f ms (median), min %.3f" % (np.median(np.array(tt2[5:])) * 1e3, np.min(np.array(tt2[5:])) * 1e3))
