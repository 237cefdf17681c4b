"""The headline configuration on a CLUTTERED scene (-m gpu): synth scene "dense" = the hall of the bench scene + 3000 thin posts,
~1 750 Gaussian cells per map -- the size SURVEY.md 8(a, d) gives a 2D map (M ~ 1-3 k), where the bench's plain hall has ~370.
A dense map exercises what the light one never does: several groups of 64 source cells per evaluation share (no hit-list
reuse across evaluations), runs of cell visits that last a few points in the build kernel, an LDS hash that is half full,
and a cell capacity that is nearly reached.

64 pairs of 100 k-point scans through the one-call entry (ndtgpu_register_batch_device): every one of the 128 maps cell by
cell against the oracle, every pose / iteration count / convergence flag against the oracle's matcher; then the same maps in a
set whose max_cells is the smallest multiple of 64 that holds them (same bits), and in one that is 64 cells too small (the
overflowing maps are flagged and their registrations refused with exit code -3, every other result unchanged)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4
POSE_TOL_RAD = 1e-4
DET_FIELDS = ["converged", "iterations", "fevals", "exit_code", "score", "n_source", "n_target", "pair_terms_g", "pair_terms_h"]
RES, SIZE, RNG = 0.5, [100.0, 100.0, 1.0], 30.0


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


def test_dense_scene_parity(N, O):
    import torch
    from ndt_feature_graph_amd import binding, synth
    dev = torch.device("cuda", 0)
    B, NP = 64, 100000
    pr = synth.pair_2d(torch.arange(1, B + 1, dtype=torch.int64, device=dev), NP, device=dev, scene="dense")
    both = torch.cat([pr["fixed"], pr["moving"]]).contiguous()
    T0 = pr["T_init"].transpose(1, 2).contiguous().reshape(B, 16)
    Ti = pr["T_init"].cpu().numpy()

    def register(max_cells):
        reg = N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=B, depth=1, max_cells=max_cells)
        T16 = T0.clone()
        res = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        reg.submit(both[:B], both[B:], T16, res, range_limit=RNG)
        reg.sync()
        ms = reg.mapset(0)
        ovf = np.array([ms.counters(k)["overflow"] for k in range(2 * B)])
        return reg, ms, T16.cpu().numpy(), res.cpu().numpy().view(binding.RESULT_DTYPE).reshape(B), ovf

    reg, ms, T16, r, ovf = register(4096)
    assert not ovf.any()
    T = T16.reshape(B, 4, 4).transpose(0, 2, 1)
    n_cells = ms.num_cells_all()
    assert 1200 < n_cells.mean() < 2600 and n_cells.max() < 4096, (n_cells.mean(), n_cells.max())     # the survey's M range
    assert np.array_equal(r["n_target"], n_cells[:B]) and np.array_equal(r["n_source"], n_cells[B:])
    assert r["converged"].mean() > 0.8 and np.all(r["exit_code"] >= 0)
    # more than 512 source cells: several groups per share, the probe of every evaluation runs (no hit-list reuse)
    assert n_cells[B:].min() > 512
    scans = both.cpu().numpy()
    omaps = []
    for k in range(2 * B):
        om = O.OracleMap(RES, [0, 0, 0], SIZE)
        om.load_points(scans[k], RNG)
        om.compute_cells()
        gm, gc, gi, gn = ms.export_cells(k)
        cm, cc, ci, cn = om.export_cells()
        assert len(gn) == len(cn) and np.array_equal(gi, ci), k                        # the same cells ...
        assert np.array_equal(gn.astype(np.int64), cn.astype(np.int64)), k             # ... holding the same points
        assert np.max(np.abs(gm - cm)) < 1e-9
        scale = np.max(np.abs(cc), axis=(1, 2), keepdims=True)
        assert np.max(np.abs(gc - cc) / scale) < 1e-8
        omaps.append(om)
    worst = 0.0
    for b in range(B):
        To, ro = O.match_d2d(omaps[b], omaps[B + b], Ti[b])
        dt = float(np.linalg.norm(T[b][:3, 3] - To[:3, 3]))
        dr = float(2.0 * np.arcsin(min(1.0, np.linalg.norm(T[b][:3, :3] - To[:3, :3]) / (2.0 * np.sqrt(2.0)))))
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (b, dt, dr)
        assert r["iterations"][b] == ro["iterations"] and bool(r["converged"][b]) == ro["converged"], b
        assert abs(r["score"][b] - ro["score"]) < 1e-6 * abs(ro["score"])
        worst = max(worst, dt)
    print("dense scene: %.0f cells per map (max %d), worst |dt| vs oracle %.2e m" % (n_cells.mean(), n_cells.max(), worst))
    reg.close()

    # a cell capacity that is nearly reached: the smallest multiple of 64 without an overflow ...
    cap = (int(n_cells.max()) + 63) // 64 * 64
    while True:
        reg2, ms2, T16b, rb, ovf2 = register(cap)
        if not ovf2.any():
            break
        reg2.close()
        cap += 64
        assert cap <= 4096
    assert np.array_equal(T16b, T16)                      # the same bits as with room to spare
    for f in DET_FIELDS:
        assert np.array_equal(rb[f], r[f]), f
    fill = max(ms2.counters(k)["n_cells"] for k in range(2 * B)) / cap
    reg2.close()
    # ... and 64 cells less: the maps that do not fit are flagged, their registrations refused, the others untouched
    reg3, ms3, T16c, rc, ovf3 = register(cap - 64)
    assert ovf3.any()
    bad = ovf3[:B].astype(bool) | ovf3[B:].astype(bool)
    assert np.all(rc["exit_code"][bad] == -3) and not rc["converged"][bad].any()
    assert np.array_equal(T16c[bad], T0.cpu().numpy()[bad])                  # the pose is left as it came
    assert np.array_equal(T16c[~bad], T16[~bad])
    for f in DET_FIELDS:
        assert np.array_equal(rc[f][~bad], r[f][~bad]), f
    print("capacity %d: fullest map %.0f %%; at %d: %d of %d registrations refused" % (cap, 100 * fill, cap - 64, int(bad.sum()), B))
    reg3.close()
