#!/usr/bin/env python3
"""3D build (64 sweeps x 200 k points, 0.25 m voxels) under the launch-shape knobs of the accumulate kernel:
NDTGPU_BUILD_XCD (bit 0: L2-local atomics, bit 1: maps dealt to XCDs), NDTGPU_BUILD_WGS.  Prints the time of each
variant and whether its cells equal the default's, bit for bit."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import synth
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
p3 = synth.pair_3d(torch.arange(1, 33, device=dev), device=dev)
sw = torch.cat([p3["fixed"], p3["moving"]]).contiguous()
m3 = N.MapSet(0.25, [0, 0, 0], [100, 100, 10], n_maps=64, max_cells=120000)
m3.profiling(True)

def digest():
    h = hashlib.sha256()
    for i in (0, 7, 13, 40, 63):
        c = m3.export_cells(i)
        for a in c:
            h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16], int(m3.num_cells_all().sum())

KNOBS = ("NDTGPU_BUILD_XCD", "NDTGPU_BUILD_WGS", "NDTGPU_FIN_WGS", "NDTGPU_RANK_WGS", "NDTGPU_PLACE_WGS")
def run(tag, **env):
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    v = []
    for _ in range(6):
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st); m3.build(sw, range_limit=70.0, stream=st); e1.record(st); torch.cuda.synchronize()
        v.append(e0.elapsed_time(e1))
    ks = [m3.last_kernel_ms(0)]
    d = digest()
    print("%-28s total %.3f ms (min %.3f) kernels %s digest %s cells %d" % (tag, float(np.median(v[1:])), min(v), ["%.3f" % x for x in ks], d[0], d[1]), flush=True)
    return d

KNOBS = ("NDTGPU_BUILD_XCD", "NDTGPU_BUILD_WGS", "NDTGPU_FIN_WGS", "NDTGPU_RANK_WGS", "NDTGPU_PLACE_WGS")
ref = run("default")
print("digest of the round-3 kernel on these sweeps: 476fd03cb5939def")
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "fin":
    for fin in (256, 512, 1024):
        run("gaussians on %d" % fin, NDTGPU_FIN_WGS=fin, NDTGPU_RANK_WGS=512, NDTGPU_PLACE_WGS=512)
    for r in (256, 1024, 2048):
        run("rank map on %d" % r, NDTGPU_RANK_WGS=r)
    for r in (256, 1024, 2048, 4096):
        run("placement on %d" % r, NDTGPU_PLACE_WGS=r)
    for a in (64, 128, 256):
        run("all three on %d" % a, NDTGPU_FIN_WGS=a, NDTGPU_RANK_WGS=a, NDTGPU_PLACE_WGS=a)
    for r in (64, 128):
        run("rank map on %d, others 256" % r, NDTGPU_FIN_WGS=256, NDTGPU_RANK_WGS=r, NDTGPU_PLACE_WGS=256)
    run("default again")
    sys.exit(0)
for fin in (1024,):
    run("fin %d" % fin, NDTGPU_FIN_WGS=fin)
for wgs in (512, 768, 1536, 2048):
    run("wgs %d" % wgs, NDTGPU_BUILD_WGS=wgs)
run("maps dealt to XCDs", NDTGPU_BUILD_XCD=2)
run("default again")
