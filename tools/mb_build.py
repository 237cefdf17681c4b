#!/usr/bin/env python3
"""Build-kernel timings: 2048 / 1024 2D scans of 100 k points in one launch, 64 3D sweeps of 200 k points (HIP events
inside the library).  NDTGPU_LIB names the library variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import synth
dev = torch.device("cuda", 0)
B = 1024
pr = synth.pair_2d(torch.arange(1, B + 1, device=dev), 100000, device=dev, chunk_bytes=2 << 30)
scans = torch.cat([pr["fixed"], pr["moving"]]).contiguous()
st = torch.cuda.current_stream()
out = []
for n in (2048, 1024):
    ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=n, max_cells=4096)
    ms.profiling(True)
    v = []
    for _ in range(6):
        ms.build(scans[:n], range_limit=30.0, stream=st)
        v.append(ms.last_kernel_ms(0))
    out.append("%d scans %.3f ms" % (n, float(np.median(v[1:]))))
    cells = ms.num_cells_all().sum()
    if "stats" in os.environ.get("NDTGPU_LIB", "") and n == 2048:
        c = np.array([ms.counters(i)["cyc"][:3] for i in range(0, n, 16)], dtype=np.float64).sum(0)
        out.append("trips all-in-run-0 %.3f, in a remembered cell %.3f, general %.3f" % (c[0] / (c[1] + c[2]), c[1] / (c[1] + c[2]), c[2] / (c[1] + c[2])))
    del ms
p3 = synth.pair_3d(torch.arange(1, 33, device=dev), device=dev)
sw = torch.cat([p3["fixed"], p3["moving"]]).contiguous()
m3 = N.MapSet(0.25, [0, 0, 0], [100, 100, 10], n_maps=64, max_cells=120000)
v = []
for _ in range(5):
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(st); m3.build(sw, range_limit=70.0, stream=st); e1.record(st); torch.cuda.synchronize()
    v.append(e0.elapsed_time(e1))
out.append("64 3D sweeps %.3f ms" % float(np.median(v[1:])))
print(os.path.basename(os.environ.get("NDTGPU_LIB", "default")), "|", " | ".join(out), "| cells", int(cells), int(m3.num_cells_all().sum()))
