#!/usr/bin/env python3
"""The flat-grid batch build kernel (csrc/ndt_build_flat.hip) against the general one (NDTGPU_FLAT=0) on the same scans:
identical cell sets / point counts, moments within rounding, rank maps, counters; HIP-event timings of both and the
flat kernel's phase clocks.  usage: python tools/flat_check.py [n_scans=2048] [points=100000] [nan]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
dev = torch.device("cuda", 0)
B = n // 2
pr = synth.pair_2d(torch.arange(1, B + 1, device=dev), npts, device=dev, chunk_bytes=2 << 30)
scans = torch.cat([pr["fixed"], pr["moving"]]).contiguous()
if len(sys.argv) > 3 and sys.argv[3] == "nan":      # a fifth of the beams without a return, scattered
    scans[torch.rand(scans.shape[:2], device=dev) < 0.2] = float("nan")
st = torch.cuda.current_stream()


def run(flat):
    os.environ["NDTGPU_FLAT"] = "2" if flat else "0"
    ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=n, max_cells=4096)
    ms.profiling(True)
    v = []
    for _ in range(6):
        ms.build(scans, range_limit=30.0, stream=st)
        v.append(ms.last_kernel_ms(0))
    return ms, float(np.median(v[1:]))


mf, tf = run(True)
mg, tg = run(False)
print("flat %.3f ms   general %.3f ms   (%d scans x %d points)" % (tf, tg, n, npts))
alg = n * (12.0 * npts)
print("flat: %.2f TB/s of points = %.1f %% of 8 TB/s" % (alg / tf / 1e9, alg / tf / 1e9 / 8 * 100))
cyc = np.array([mf.counters(i)["cyc"][:3] for i in range(0, n, max(1, n // 64))], dtype=np.float64)
print("flat phase clocks (mean over sampled maps): A %.0f  B %.0f  C %.0f" % tuple(cyc.mean(0)))
if "stats" in os.environ.get("NDTGPU_LIB", ""):
    c4 = np.array([mf.counters(i)["cyc"] for i in range(0, n, max(1, n // 64))], dtype=np.float64).mean(0)
    print("flat stats per map: rounds with the index arithmetic %.0f, rounds that failed the fast tests %.0f, flushes %.0f, drains %.0f (of %d rounds)" % (c4[0], c4[1], c4[2], c4[3], (npts + 63) // 64))
bad = 0
worst = 0.0
for i in list(range(0, n, max(1, n // 48))) + [n - 1]:
    a, b = (dict(zip(("mean", "cov", "idx", "npts"), m.export_cells(i))) for m in (mf, mg))
    cf, cg = mf.counters(i), mg.counters(i)
    same = a["idx"].shape == b["idx"].shape and np.array_equal(a["idx"], b["idx"]) and np.array_equal(a["npts"], b["npts"])
    if not same or cf["n_dropped"] != cg["n_dropped"] or cf["overflow"] != cg["overflow"]:
        bad += 1
        print("map", i, "differs: cells", a["idx"].shape[0], b["idx"].shape[0], "dropped", cf["n_dropped"], cg["n_dropped"])
        sa = {tuple(r): k for r, k in zip(a["idx"].tolist(), a["npts"].tolist())}
        sb = {tuple(r): k for r, k in zip(b["idx"].tolist(), b["npts"].tolist())}
        print("   only flat:", [(k, v) for k, v in sa.items() if k not in sb][:6], " only general:", [(k, v) for k, v in sb.items() if k not in sa][:6],
              " other n:", [(k, sa[k], sb[k]) for k in sa if k in sb and sa[k] != sb[k]][:6])
        continue
    if a["mean"].size:
        worst = max(worst, float(np.abs(a["mean"] - b["mean"]).max()), float(np.abs(a["cov"] - b["cov"]).max()))
print("sampled maps: %d differ; max |mean/cov difference| %.3e; cells/map %.1f" % (bad, worst, mf.num_cells_all().mean()))
# run-to-run identical bits
m2, _ = run(True)
i = n // 3
a, b = mf.export_cells(i), m2.export_cells(i)
print("flat run-to-run identical:", bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])))
