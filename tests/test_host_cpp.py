"""The C++ host mirror (ndt_feature_graph_amd/host/*.h: lslgeneric::NDTMap / NDTMatcherD2D[_2D] and
ndt_feature::NDTFeatureFuserHMT / NDTFeatureGraph over the C-ABI) compiles with plain g++ and behaves: without a GPU it
fails loudly; with a GPU tests/native/host_demo.cpp drives the graph front door over a trajectory, refines all links in batched
calls, and runs the reference's Newton loop (ndt_matcher_d2d_fusion.h:847-1121) RE-TYPED against the mirror -- one
ndtgpu_derivatives call per evaluation -- landing on the device-resident matchFusion's pose."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "native")       # host_demo.cpp: the harness; the mirror it tests is ndt_feature_graph_amd/host/*.h


def _build():
    import ndt_feature_graph_amd as N
    N.build_library()
    subprocess.check_call(["make", "-s", "-C", HOST])
    return os.path.join(HOST, "host_demo")


def test_host_demo_without_gpu_fails_loudly():
    import ndt_feature_graph_amd as N
    exe = _build()
    if N.device_count() > 0:
        pytest.skip("a GPU is present")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "no CPU fallback" in out.stdout


@pytest.mark.gpu
def test_host_demo_on_gpu():
    exe = _build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failures in total" in out.stdout and "batch==single 1" in out.stdout
    assert "A: " in out.stdout and "C: soft=1" in out.stdout and "E: message" in out.stdout
