#!/usr/bin/env python3
"""Build time of n 2D scans (100 k points) for n below the batch threshold: the general split path (NDTGPU_FLAT=1: the
flat kernel only from 256 maps on) against the flat kernel with one workgroup per map (NDTGPU_FLAT=2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import synth
dev = torch.device("cuda", 0)
pr = synth.pair_2d(torch.arange(1, 257, device=dev), 100000, device=dev)
scans = torch.cat([pr["fixed"], pr["moving"]]).contiguous()
st = torch.cuda.current_stream()
for n in (32, 64, 96, 128, 192, 255, 256, 512):
    out = []
    for flat in ("1", "2"):
        os.environ["NDTGPU_FLAT"] = flat
        ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=n, max_cells=4096)
        v = []
        for _ in range(6):
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(st); ms.build(scans[:n], range_limit=30.0, stream=st); e1.record(st); torch.cuda.synchronize()
            v.append(e0.elapsed_time(e1))
        out.append("%.3f" % float(np.median(v[1:])))
        cells = int(ms.num_cells_all().sum())
        del ms
    print("%4d scans: general/split %s ms, flat %s ms (cells %d)" % (n, out[0], out[1], cells), flush=True)
