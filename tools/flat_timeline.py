#!/usr/bin/env python3
"""Timeline of one launch of the flat build kernel (library built with -DNDT_FLAT_TIMES: every workgroup records its start
and end on the 100 MHz clock, its core clocks and where it ran).  usage: NDTGPU_LIB=.../libndtgpu_times.so python tools/flat_timeline.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import synth

n, npts = 2048, 100000
dev = torch.device("cuda", 0)
pr = synth.pair_2d(torch.arange(1, n // 2 + 1, device=dev), npts, device=dev, chunk_bytes=2 << 30)
scans = torch.cat([pr["fixed"], pr["moving"]]).contiguous()
st = torch.cuda.current_stream()
os.environ["NDTGPU_FLAT"] = "2"
ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=n, max_cells=4096)
ms.profiling(True)
for _ in range(4):
    ms.build(scans, range_limit=30.0, stream=st)
print("launch %.3f ms (HIP events)" % ms.last_kernel_ms(0))
c = np.array([ms.counters(i)["cyc"] for i in range(n)], dtype=np.int64)
t0, t1, clk, where = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
base = t0.min()
s, e = (t0 - base) / 100.0, (t1 - base) / 100.0          # microseconds
dur = e - s
print("workgroups: first start 0, last start %.1f us, last end %.1f us" % (s.max(), e.max()))
print("duration per workgroup: mean %.1f us  min %.1f  max %.1f" % (dur.mean(), dur.min(), dur.max()))
print("phase A, clocks of the slowest wave / mean of the four waves: %.0f / %.0f  (mean over maps: the waves wait %.1f %% of phase A for the slowest)" % (clk.mean(), where.mean(), 100.0 * (1.0 - where.mean() / clk.mean())))
order = np.argsort(s)
print("start times (us) of workgroups in start order, every 128th:", np.round(s[order][::128], 1))
print("end times, every 128th:", np.round(np.sort(e)[::128], 1))
# concurrency: how many workgroups are running at time t
ts = np.linspace(0, e.max(), 21)
print("running workgroups at t:", [(round(float(t)), int(((s <= t) & (e > t)).sum())) for t in ts])
