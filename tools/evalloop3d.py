#!/usr/bin/env python3
"""Shader clocks per derivative evaluation of one 3D pair (12 k-cell maps) on ONE workgroup, and their sections (library
built with -DNDT_MATCH_PROF; NDTGPU_LIB names it): 24 passes of 512 source cells, i.e. 24 of the task pool's four-chunk
tasks without their hand-over."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import binding, synth
dev = torch.device("cuda", 0)
pr = synth.pair_3d(torch.tensor([1], device=dev), device=dev)
ms = N.MapSet(0.25, [0, 0, 0], [100.0, 100.0, 10.0], n_maps=2, max_cells=120000)
ms.build(torch.cat([pr["fixed"], pr["moving"]]).contiguous(), range_limit=70.0)
torch.cuda.synchronize()
L = binding.lib()
T = np.ascontiguousarray(pr["T_init"][0].cpu().numpy().T.reshape(-1))
out = np.zeros(4, dtype=np.int64)
L.ndtgpu_debug_eval_loop.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_double, C.c_double, C.POINTER(C.c_longlong)]
L.ndtgpu_debug_prof.argtypes = [C.POINTER(C.c_longlong), C.c_int]
P = binding.MatchParams(); L.ndtgpu_default_match_params(C.byref(P))
n_it = 20
prof = np.zeros(16, dtype=np.int64)
wavep = np.zeros(16, dtype=np.int64)
L.ndtgpu_debug_wave_prof.argtypes = [C.POINTER(C.c_longlong), C.c_int]
for with_h in (0, 1):
    L.ndtgpu_debug_prof(prof.ctypes.data_as(C.POINTER(C.c_longlong)), 1)
    L.ndtgpu_debug_wave_prof(wavep.ctypes.data_as(C.POINTER(C.c_longlong)), 1)
    rc = L.ndtgpu_debug_eval_loop(ms.h, 0, ms.h, 1, T.ctypes.data_as(C.POINTER(C.c_double)), with_h, n_it, 0, 1, P.lfd1, P.lfd2,
                                  out.ctypes.data_as(C.POINTER(C.c_longlong)))
    assert rc == 0, rc
    L.ndtgpu_debug_prof(prof.ctypes.data_as(C.POINTER(C.c_longlong)), 0)
    L.ndtgpu_debug_wave_prof(wavep.ctypes.data_as(C.POINTER(C.c_longlong)), 0)
    print("   per share: k clocks until the final barrier", np.round(wavep[:8] / n_it / 1e3).astype(int), " k pair terms", np.round(wavep[8:] / n_it / 1e3, 1))
    p = prof[8:14] if with_h else prof[0:6]
    per = out[0] / n_it
    passes = (out[2] + 511) // 512
    print("with_h %d: %.0f k clocks per evaluation of %d cells (%d pair terms) = %.1f k per pass of 512 cells; sections of wave 0 per pass: "
          "transform %.1f k, probe + count %.1f k, fill %.1f k, terms %.1f k, rest %.1f k" % (
              with_h, per / 1e3, out[2], out[1], per / passes / 1e3, *(p[k] / max(1, p[5]) / passes / 1e3 for k in range(5))))
