#!/usr/bin/env python3
"""Flat-grid batch build kernel against the general one on the DENSE synthetic scene (~1.7 k cells per map).
usage: python tools/dense_build_ab.py [pairs=128]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
pr = synth.pair_2d(torch.arange(1, B + 1, device=dev), 100000, device=dev, chunk_bytes=1 << 30, scene="dense")
scans = torch.cat([pr["fixed"], pr["moving"]]).contiguous()
for flat in ("2", "0"):
    os.environ["NDTGPU_FLAT"] = flat
    ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=2 * B, max_cells=4096)
    ms.profiling(True)
    v = []
    for _ in range(6):
        ms.build(scans, range_limit=30.0, stream=torch.cuda.current_stream())
        v.append(ms.last_kernel_ms(0))
    print("NDTGPU_FLAT=%s: %d scans %.3f ms, cells/map %.0f" % (flat, 2 * B, float(np.median(v[1:])), ms.num_cells_all().mean()))
