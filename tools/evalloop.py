#!/usr/bin/env python3
"""Shader clocks per derivative evaluation of one 2D pair on ONE workgroup (library built with -DNDT_MATCH_PROF and,
for the experiments, -DNDT_EXP_*; NDTGPU_LIB names it)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import binding, synth
pr = synth.pair_2d([1, 2], 100000)
ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=4, max_cells=4096)
ms.build(np.concatenate([pr["fixed"].numpy(), pr["moving"].numpy()]), range_limit=30.0)
L = binding.lib()
T = np.ascontiguousarray(pr["T_init"][0].numpy().T.reshape(-1))
out = np.zeros(4 * 256, dtype=np.int64)
L.ndtgpu_debug_eval_loop.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_double, C.c_double, C.POINTER(C.c_longlong)]
P = binding.MatchParams(); L.ndtgpu_default_match_params(C.byref(P))
n_it = 200
print(os.environ.get("NDTGPU_LIB", "default"))
for groups in (1, 256):
    for with_h in (0, 1):
        for reuse in (1, 0):
            for rep in range(2):
                rc = L.ndtgpu_debug_eval_loop(ms.h, 0, ms.h, 2, T.ctypes.data_as(C.POINTER(C.c_double)), with_h, n_it, reuse, groups, P.lfd1, P.lfd2,
                                              out.ctypes.data_as(C.POINTER(C.c_longlong)))
                assert rc == 0, rc
            o = out[:4 * groups].reshape(groups, 4)
            print("  groups %3d  with_h %d  reuse %d : %7.0f clocks per evaluation (%d pair terms, %d source cells)" % (
                groups, with_h, reuse, o[:, 0].mean() / n_it, o[0, 1], o[0, 2]))
