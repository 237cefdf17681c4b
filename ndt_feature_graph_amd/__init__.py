"""ndt_feature_graph_amd -- MI355X-native NDT scan-matching front-end (grid build + D2D matcher)
behind the lslgeneric::NDTMap / NDTMatcherD2D call sites of MalcolmMielle/ndt_feature_graph.

The product is the C-ABI shared library (include/ndtgpu.h, csrc/*.hip -> libndtgpu.so).  This
Python package is only the thin ctypes host layer used by tests/ and bench.py; the C++ host
mirror of the reference classes lives in ndt_feature_graph_amd/host/.

There is no CPU fallback: importing works anywhere (the build check runs without a GPU), but
every compute call raises NdtGpuError when the HIP library or a device is missing.
"""
from .binding import (MapSet, Registrar, FuserBank, fuser_params, fuser_prepare, covariance, MatchParams, NdtGpuError, build_library, derivatives, device_count, lib,  # noqa: F401
                      library_path, match_batch, match_d2d, match_fusion_batch, overlap_score)
