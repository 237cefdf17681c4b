import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import binding, synth
dev = torch.device("cuda", 0)
B = 1024
pr = synth.pair_2d(torch.arange(1, B + 1, device=dev), 100000, device=dev, chunk_bytes=2 << 30)
both = torch.cat([pr["fixed"], pr["moving"]]).contiguous()
ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=2 * B, max_cells=4096)
st = torch.cuda.current_stream()
ms.build(both, range_limit=30.0, stream=st)
Ti = pr["T_init"].transpose(1, 2).contiguous().reshape(B, 16)
T16 = Ti.clone(); res = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
ti = torch.arange(B, dtype=torch.int32, device=dev); si = ti + B
for mask in (0x3f, 0x23):
    ts = []
    for _ in range(6):
        T16.copy_(Ti); torch.cuda.synchronize(); t0 = time.perf_counter()
        binding.match_batch_device(ms, ti, ms, si, T16, res, B, stream=st, dof_mask=mask); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    r = res.cpu().numpy().view(binding.RESULT_DTYPE).reshape(B)
    print("dof_mask 0x%x: %.3f ms per 1024 pairs (min of 5), iterations %.2f, fevals %.1f, pair terms g %.1f M h %.1f M, converged %.2f" % (
        mask, 1e3 * min(ts[1:]), r["iterations"].mean(), r["fevals"].mean(), r["pair_terms_g"].sum() / 1e6, r["pair_terms_h"].sum() / 1e6, r["converged"].mean()))
