#!/usr/bin/env python3
"""Section clocks of the ranking launch (MODE 3) of the 3D build; library built with -DNDT_BUILD_PROF."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import synth
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
p3 = synth.pair_3d(torch.arange(1, 33, device=dev), device=dev)
sw = torch.cat([p3["fixed"], p3["moving"]]).contiguous()
m3 = N.MapSet(0.25, [0, 0, 0], [100, 100, 10], n_maps=64, max_cells=120000)
for fin in (512, 1024, 2048):
    os.environ["NDTGPU_FIN_WGS"] = str(fin)
    for rep in range(3):
        c0 = np.array([m3.counters(i)["cyc"] for i in range(64)], dtype=np.float64)
        m3.build(sw, range_limit=70.0, stream=st); torch.cuda.synchronize()
        c1 = np.array([m3.counters(i)["cyc"] for i in range(64)], dtype=np.float64)
    d = (c1 - c0) * 16
    wg = fin // 64
    print("fin %d (%d workgroups per map): per workgroup (clocks): to the end of the look-back %.0f, pass 2 + gather %.0f; the last one's clean-up %.0f"
          % (fin, wg, d[:, 0].mean() / wg, d[:, 1].mean() / wg, d[:, 2].mean()))
