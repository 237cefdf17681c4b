"""NDTMatcherD2D::covariance on the device (ndt_feature_graph.cpp:296-298, ndt_feature_fuser_hmt.cpp:403-405) vs the
CPU oracle and vs an independent NumPy restatement of cov = H^-1 (0.03^2 J^T J) H^-1 built on the device Hessian
(which tests/test_gpu_parity.py validates against finite differences).  -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    import ndt_feature_graph_amd as N
    if N.device_count() < 1:
        pytest.fail("no HIP device visible: the HIP path cannot run (there is no CPU fallback)")
    return N


@pytest.fixture(scope="module")
def O():
    import oracle
    return oracle


def _cross(k):
    E = np.zeros((3, 3))
    a, b = (k + 1) % 3, (k + 2) % 3
    E[a, b], E[b, a] = -1.0, 1.0
    return E


def cov_numpy(N, ms, ti, si, T, mode, lfd1=1.0, lfd2=0.05, nn=2):
    """J rows in NumPy from exported cells, H from the device derivatives at T."""
    tm, tc, tidx, _ = ms.export_cells(ti)
    sm, sc, _, _ = ms.export_cells(si)
    info = ms.info()
    size = np.array(info["cells_per_axis"])
    R, t = T[:3, :3], T[:3, 3]
    m = sm @ R.T + t
    C = R @ sc @ R.T
    _, _, H = N.derivatives(ms, ti, m, C, n_neighbours=nn, lfd1=lfd1, lfd2=lfd2)
    lut = {tuple(i): k for k, i in enumerate(tidx)}
    sig = 0.03 ** 2
    JK = np.zeros((6, 6))
    for k in range(len(m)):
        idx = tuple((np.floor(m[k] / ms.res + 0.5) + size / 2.0).astype(int))
        j = lut.get(idx)
        if j is None:
            continue
        x = m[k] - tm[j]
        S = tc[j] + C[k]
        if abs(np.linalg.det(S)) <= 1e-12:
            continue
        B = np.linalg.inv(S)
        f = -x @ B @ x / 2
        if f < -120:
            continue
        f = np.exp(lfd2 * f) / 2
        Q = -sig * B @ B
        G = np.zeros(6)
        G[:3] = x @ Q
        if mode == 0:
            for a in range(3):
                E = _cross(a)
                Z = E @ C[k] + C[k] @ E.T
                G[3 + a] = x @ Q @ (E @ m[k]) - x @ Q @ Z @ B @ x - x @ B @ Z @ Q @ x
        G = (G + (-lfd2 / 2) * (x @ Q @ x)) * f * lfd1 * lfd2 / 2
        JK += np.outer(G, G)
    Hi = np.linalg.inv(H)
    return Hi @ (sig * JK) @ Hi


def test_covariance_parity(N, O):
    from ndt_feature_graph_amd import synth
    seeds = [1, 2, 3, 4, 5, 6]
    B = len(seeds)
    pr = synth.pair_2d(seeds, 30000)
    ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=2 * B)
    ms.build(np.concatenate([pr["fixed"].numpy(), pr["moving"].numpy()]), range_limit=30.0)
    T0 = pr["T_init"].numpy()
    T, r = N.match_batch(ms, np.arange(B), ms, np.arange(B) + B, T0)
    for mode in (0, 1):
        cov, sing = N.covariance(ms, np.arange(B), ms, np.arange(B) + B, T, mode=mode)
        assert not sing.any()
        for b in range(B):
            ot = O.OracleMap(0.5, [0, 0, 0], [100, 100, 1]); ot.load_points(pr["fixed"][b].numpy(), 30.0); ot.compute_cells()
            os_ = O.OracleMap(0.5, [0, 0, 0], [100, 100, 1]); os_.load_points(pr["moving"][b].numpy(), 30.0); os_.compute_cells()
            co = O.covariance(ot, os_, T[b], mode=mode)
            scale = np.abs(co).max()
            assert scale > 0
            assert np.max(np.abs(cov[b] - co)) < 1e-7 * scale, (mode, b, np.max(np.abs(cov[b] - co)) / scale)
            assert np.max(np.abs(cov[b] - cov[b].T)) < 1e-9 * scale
            assert np.linalg.eigvalsh(0.5 * (cov[b] + cov[b].T)).min() > -1e-9 * scale        # H^-1 (J^T J) H^-1 is PSD
            if b < 2:
                cn = cov_numpy(N, ms, b, b + B, T[b], mode)
                assert np.max(np.abs(cov[b] - cn)) < 1e-7 * scale, (mode, b)
    # a single link through the same entry gives the same bits as the batch
    c1, _ = N.covariance(ms, [2], ms, [2 + B], T[2:3], mode=0)
    ca, _ = N.covariance(ms, np.arange(B), ms, np.arange(B) + B, T, mode=0)
    assert np.array_equal(c1[0], ca[2])
    # empty source map: H = 0 is singular -> flagged, all-zero matrix, no error
    ms2 = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=2)
    ms2.build(np.stack([pr["fixed"][0].numpy(), np.full_like(pr["fixed"][0].numpy(), np.nan)]), range_limit=30.0)
    c0, s0 = N.covariance(ms2, [0], ms2, [1], np.eye(4)[None])
    assert s0[0] == 1 and not c0.any()
