"""Host-side checks of csrc/ndt_math.h (the source the device compiles): lambda_min / lambda_max by Householder
tridiagonalisation + Laguerre against numpy.linalg.eigvalsh and against the cyclic Jacobi it replaces in the Newton
loop's regulariser (fusion.h:922-940).  Only host code runs: no GPU needed."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "math_harness.hip")
EXE = os.path.join(ROOT, "tests", "native", "math_harness")


@pytest.fixture(scope="module")
def harness():
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    deps = [SRC, os.path.join(ROOT, "ndt_feature_graph_amd", "csrc", "ndt_math.h")]
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", SRC, "-o", EXE])

    def run(mats, mode=None):
        mats = np.ascontiguousarray(mats, dtype=np.float64)
        n = mats.shape[0]
        out = subprocess.run([EXE] + ([mode] if mode else []), input=np.uint32(n).tobytes() + mats.tobytes(),
                             stdout=subprocess.PIPE, check=True).stdout
        return np.frombuffer(out, dtype=np.float64).reshape(n, -1)
    return run


def _cases():
    g = np.random.default_rng(2)
    mats = []
    for _ in range(400):                                   # generic indefinite matrices over 12 decades of scale
        A = g.normal(size=(6, 6)) * 10.0 ** g.uniform(-4, 8)
        mats.append(A + A.T)
    for _ in range(200):                                   # Hessian-like: stiff rotation block, soft translation block
        A = g.normal(size=(6, 6))
        S = np.diag([1, 1, 1e-3, 30, 30, 300.0])
        mats.append(S @ (A + A.T) @ S * 10.0 ** g.uniform(0, 4))
    for _ in range(100):                                   # prescribed spectra: clusters, multiple and zero eigenvalues
        Q, _ = np.linalg.qr(g.normal(size=(6, 6)))
        ev = g.choice([-3.0, -3.0, 1e-9, 0.0, 2.0, 2.0, 2.0 + 1e-12, 5.0, 1e6], 6)
        mats.append(Q @ np.diag(ev) @ Q.T)
    mats.append(np.zeros((6, 6)))
    mats.append(np.eye(6) * 7.5)
    mats.append(np.diag([1.0, -2.0, 3.0, -4.0, 5.0, 6.0]))
    T = np.diag([2.0] * 6) + np.diag([-1.0] * 5, 1) + np.diag([-1.0] * 5, -1)      # already tridiagonal
    mats.append(T)
    return np.stack(mats)


def test_sym6_extreme_eigs_against_numpy_and_jacobi(harness):
    mats = _cases()
    out = harness(mats)
    ev = np.linalg.eigvalsh(mats)
    norm = np.maximum(np.abs(ev).max(axis=1), 1e-300)
    err_lo = np.abs(out[:, 0] - ev[:, 0]) / norm
    err_hi = np.abs(out[:, 1] - ev[:, -1]) / norm
    jac_lo = np.abs(out[:, 2] - ev[:, 0]) / norm
    jac_hi = np.abs(out[:, 3] - ev[:, -1]) / norm
    assert err_lo.max() < 5e-14 and err_hi.max() < 5e-14, (err_lo.max(), err_hi.max())
    assert jac_lo.max() < 5e-14 and jac_hi.max() < 5e-14
    # the sign decision of the regulariser (lambda_min < 0) agrees wherever lambda_min is not rounding noise
    clear = np.abs(ev[:, 0]) > 1e-12 * norm
    assert np.array_equal(out[clear, 0] < 0, ev[clear, 0] < 0)


def test_sincos_pose_small_angle_kernels(harness):
    """pose_to_rigid's sin / cos (fdlibm kernels up to pi/4, library beyond) against numpy: <= 1 ulp."""
    g = np.random.default_rng(3)
    x = np.concatenate([g.uniform(-np.pi / 4, np.pi / 4, 20000), g.uniform(-1e-3, 1e-3, 2000), g.uniform(-4, 4, 2000),
                        [0.0, np.pi / 4, -np.pi / 4, 0.7853981633974484, 1e-300, -1e-9, 3.0]])
    out = harness(x, "s")
    s, c = np.sin(x), np.cos(x)
    assert np.max(np.abs(out[:, 0] - s) / np.maximum(np.spacing(np.abs(s)), 1e-320)) <= 1.0
    assert np.max(np.abs(out[:, 1] - c) / np.spacing(np.abs(c))) <= 1.0
