#!/usr/bin/env python3
"""Where the 3D batch matcher (32 pairs of 12 k-cell maps, grid-barrier kernel) spends its time: per registration the
shader clocks of workgroup 0 in evaluations and in the solver, against the launch time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import binding, synth
dev = torch.device("cuda", 0)
B = int(os.environ.get("PAIRS", "32"))
pr = synth.pair_3d(torch.arange(1, B + 1, device=dev), device=dev)
sweeps = torch.cat([pr["fixed"], pr["moving"]]).contiguous()
ms = N.MapSet(0.25, [0, 0, 0], [100.0, 100.0, 10.0], n_maps=2 * B, max_cells=120000)
st = torch.cuda.current_stream()
ms.build(sweeps, range_limit=70.0, stream=st)
Ti = pr["T_init"].transpose(1, 2).contiguous().reshape(B, 16)
T16 = Ti.clone()
results = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
ti = torch.arange(B, dtype=torch.int32, device=dev); si = ti + B
ts = []
for _ in range(5):
    T16.copy_(Ti); torch.cuda.synchronize(); t0 = time.perf_counter()
    binding.match_batch_device(ms, ti, ms, si, T16, results, B, stream=st); torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
r = results.cpu().numpy().view(binding.RESULT_DTYPE).reshape(B)
ms_launch = 1e3 * min(ts[1:])
print("%d pairs: %.2f ms; iterations %.1f, evaluations %.1f, cells %d; workgroup 0: evaluations %.2f ms, solver %.2f ms (at 2.4 GHz); pair terms g %.0f k, h %.0f k per evaluation" % (
    B, ms_launch, r["iterations"].mean(), r["fevals"].mean(), r["n_source"].mean(), r["cycles_eval"].mean() / 2.4e6, r["cycles_solver"].mean() / 2.4e6,
    r["pair_terms_g"].sum() / max(1, (r["fevals"] - r["iterations"]).sum()) / 1e3, r["pair_terms_h"].sum() / max(1, r["iterations"].sum()) / 1e3))
o = np.argsort(-r["fevals"])
print("per pair (by evaluations): evaluations", r["fevals"][o][:8], "iterations", r["iterations"][o][:8])
print("   workgroup-0 evaluation ms", np.round(r["cycles_eval"][o][:8] / 2.4e6, 2), " per evaluation us", np.round(r["cycles_eval"][o][:8] / 2.4e3 / r["fevals"][o][:8], 1))
