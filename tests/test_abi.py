"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every
symbol include/ndtgpu.h declares, and fails LOUDLY (no CPU fallback) when no device is present."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def N():
    import ndt_feature_graph_amd as N
    N.build_library()
    return N


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ndtgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ndtgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(N):
    from ndt_feature_graph_amd import binding
    L = ctypes.CDLL(N.library_path())
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), "libndtgpu.so does not export %s" % s
    assert sorted(binding.EXPORTS) == syms, "binding.EXPORTS out of sync with include/ndtgpu.h"


def test_header_cites_reference_for_each_entry_point():
    text = open(os.path.join(ROOT, "include", "ndtgpu.h")).read()
    for needle in ("ndt_feature_graph.cpp:273", "ndt_feature_fuser_hmt.cpp", "ndt_matcher_d2d_fusion.h",
                   "ndt_feature_graph.cpp:347-353"):
        assert needle in text


def test_struct_layouts_match_header(N):
    from ndt_feature_graph_amd import binding
    assert ctypes.sizeof(binding.MatchResult) == 64
    assert ctypes.sizeof(binding.MatchParams) == 48
    assert ctypes.sizeof(binding.GridParams) == 64
    assert ctypes.sizeof(binding.CellParams) == 16
    p = binding.match_params()
    assert (p.n_neighbours, p.itr_max, p.step_control, p.dof_mask, p.use_initial_guess) == (2, 30, 1, 0x3F, 1)
    assert p.delta_score == 1e-6 and p.lfd1 == 1.0 and p.lfd2 == 0.05


def test_struct_sizes_against_the_compiled_header(N, tmp_path):
    """sizeof of every struct that crosses the boundary, as gcc sees include/ndtgpu.h, against the ctypes mirrors"""
    import subprocess
    from ndt_feature_graph_amd import binding
    names = {"ndtgpu_match_result": binding.MatchResult, "ndtgpu_match_params": binding.MatchParams,
             "ndtgpu_grid_params": binding.GridParams, "ndtgpu_cell_params": binding.CellParams,
             "ndtgpu_registrar_params": binding.RegistrarParams, "ndtgpu_registrar_info": binding.RegistrarInfo,
             "ndtgpu_fuser_params": binding.FuserParams, "ndtgpu_fuser_prepared": binding.FuserPrepared}
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "ndtgpu.h"\nint main(void) {\n' +
                   "".join('  printf("%s %%zu\\n", sizeof(%s));\n' % (n, n) for n in names) +
                   '  printf("ndtgpu_fuser_result %zu\\n", sizeof(ndtgpu_fuser_result));\n  return 0;\n}\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    for n, c in names.items():
        assert int(out[n]) == ctypes.sizeof(c), (n, out[n], ctypes.sizeof(c))
    assert int(out["ndtgpu_fuser_result"]) == binding.FUSER_RESULT_DTYPE.itemsize == 560


def test_no_device_fails_loudly_no_cpu_fallback(N):
    if N.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(N.NdtGpuError) as e:
        N.MapSet(0.5, [0, 0, 0], [100, 100, 1])
    assert e.value.status == -3          # NDTGPU_ERR_NO_DEVICE
    # the product never imports the oracle
    import ndt_feature_graph_amd
    pkg = os.path.dirname(ndt_feature_graph_amd.__file__)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in src and "ndt_oracle" not in src, f


def test_kernel_names(N):
    L = N.lib()
    assert L.ndtgpu_kernel_name(0) == b"ndt_build_kernel"
    assert L.ndtgpu_kernel_name(1) == b"ndt_match_kernel"
