#!/usr/bin/env python3
"""Section clocks of the 3D accumulate launch (library built with -DNDT_BUILD_PROF, NDTGPU_LIB names it): wave 0 of every
workgroup adds its clocks / 16 to the map's counters; the finalise launches are skipped (NDTGPU_BUILD_XCD=16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import synth
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
p3 = synth.pair_3d(torch.arange(1, 33, device=dev), device=dev)
sw = torch.cat([p3["fixed"], p3["moving"]]).contiguous()
for wgs, xcd in ((768, 16), (768, 16), (768, 20), (768, 24), (768, 16), (768, 24)):       # (the first one warms the process up; 20: no accumulator atomics)
    os.environ["NDTGPU_BUILD_XCD"] = str(xcd)
    os.environ["NDTGPU_BUILD_WGS"] = str(wgs)
    m3 = N.MapSet(0.25, [0, 0, 0], [100, 100, 10], n_maps=64, max_cells=120000)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(st); m3.build(sw, range_limit=70.0, stream=st); e1.record(st); torch.cuda.synchronize()
    c = np.array([m3.counters(i)["cyc"] for i in range(64)], dtype=np.float64).sum(0) * 16 / wgs
    print("wgs %d xcd %d: launch %.3f ms; per wave (clocks): staging %.0f, point loop %.0f (of which drains %.0f), end of super-tile %.0f (drains in both: %.0f)"
          % (wgs, xcd, e0.elapsed_time(e1), c[0], c[1], 0, c[3], c[2]), "alloc", m3.counters(0)["n_alloc"])
    del m3
