"""ctypes binding of oracle/libndt_oracle.so (plain-C fp64 restatement of the reference path).

Test infrastructure only.  Never imported by ndt_feature_graph_amd (the product).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libndt_oracle.so")


def build(force=False):
    """gcc-compile the C restatement (seconds)."""
    src = os.path.join(_HERE, "ndt_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libndt_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


class MatchParams(C.Structure):
    _fields_ = [("n_neighbours", C.c_int), ("itr_max", C.c_int), ("delta_score", C.c_double),
                ("step_control", C.c_int), ("lfd1", C.c_double), ("lfd2", C.c_double),
                ("dof_mask", C.c_int), ("use_initial_guess", C.c_int)]


class MatchResult(C.Structure):
    _fields_ = [("converged", C.c_int), ("iterations", C.c_int), ("fevals", C.c_int),
                ("score", C.c_double), ("exit_code", C.c_int)]


PHI_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_double, C.POINTER(C.c_double))

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int)
    L.oracle_map_create.restype = C.c_void_p
    L.oracle_map_create.argtypes = [C.c_double, dp, dp]
    L.oracle_map_destroy.argtypes = [C.c_void_p]
    L.oracle_map_index_for_point.argtypes = [C.c_void_p, dp, ip]
    L.oracle_map_load_points.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_double, dp]
    L.oracle_map_compute_cells.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.oracle_map_compute_cells_full.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
    L.oracle_map_add_point_cloud.argtypes = [C.c_void_p, dp, C.c_void_p, C.c_size_t, C.c_size_t, C.c_double, C.c_double,
                                             C.c_double, C.c_int]
    L.oracle_beam_evidence.argtypes = [dp, dp, dp, C.POINTER(C.c_float), C.c_double, C.POINTER(C.c_float)]
    L.oracle_map_occupancy.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.oracle_map_size.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.oracle_occupancy_rescaled.restype = C.c_float
    L.oracle_occupancy_rescaled.argtypes = [C.c_float]
    L.oracle_overlap_score.restype = C.c_double
    L.oracle_overlap_score.argtypes = [C.c_void_p, C.c_void_p, dp, C.POINTER(C.c_longlong)]
    L.oracle_map_num_cells.argtypes = [C.c_void_p]
    L.oracle_map_export_cells.argtypes = [C.c_void_p, dp, dp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.oracle_map_set_cells.argtypes = [C.c_void_p, dp, dp, C.c_size_t]
    L.oracle_derivatives.restype = C.c_double
    L.oracle_derivatives.argtypes = [C.c_void_p, dp, dp, C.c_size_t, C.c_int, C.c_int, C.c_double,
                                     C.c_double, dp, dp]
    L.oracle_score_at.restype = C.c_double
    L.oracle_score_at.argtypes = [C.c_void_p, C.c_void_p, dp, dp, C.c_int, C.c_double, C.c_double]
    L.oracle_match_d2d.argtypes = [C.c_void_p, C.c_void_p, dp, C.POINTER(MatchParams), C.POINTER(MatchResult)]
    L.oracle_match_fusion.argtypes = [C.c_void_p, C.c_void_p, dp, C.POINTER(MatchParams), dp, C.c_int, C.POINTER(MatchResult)]
    L.oracle_covariance.argtypes = [C.c_void_p, C.c_void_p, dp, C.c_int, C.c_double, C.c_double, C.c_int, dp]
    L.oracle_mt_cstep.argtypes = [dp, dp, dp, dp, dp, dp, dp, C.c_double, C.c_double, ip, C.c_double, C.c_double]
    L.oracle_mt_linesearch.restype = C.c_double
    L.oracle_mt_linesearch.argtypes = [PHI_FN, C.c_void_p, C.c_double, C.c_double, ip, ip]
    L.oracle_pose_to_T.argtypes = [dp, dp]
    L.oracle_eig_sym.argtypes = [C.c_int, dp, dp, dp]
    L.oracle_ldlt_solve.argtypes = [C.c_int, dp, dp, dp]
    L.oracle_mahalanobis.restype = C.c_double
    L.oracle_mahalanobis.argtypes = [dp, dp, dp, dp]
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


DEFAULT_PARAMS = dict(n_neighbours=2, itr_max=30, delta_score=1e-6, step_control=1, lfd1=1.0, lfd2=0.05,
                      dof_mask=0x3F, use_initial_guess=1)


class OracleMap:
    """lslgeneric::NDTMap(new LazyGrid(res)) restated on the CPU."""

    def __init__(self, res, centre, size_m):
        self._L = lib()
        c = _f64(centre)
        s = _f64(size_m)
        self.h = self._L.oracle_map_create(float(res), _dp(c), _dp(s))
        if not self.h:
            raise MemoryError("oracle_map_create")

    def __del__(self):
        if getattr(self, "h", None):
            self._L.oracle_map_destroy(self.h)
            self.h = None

    def index_for_point(self, p):
        p = _f64(p)
        idx = (C.c_int * 3)()
        inside = self._L.oracle_map_index_for_point(self.h, _dp(p), idx)
        return list(idx), bool(inside)

    def load_points(self, xyz, range_limit=-1.0, range_origin=None):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        assert xyz.ndim == 2 and xyz.shape[1] in (3, 4)
        ro = _f64(range_origin) if range_origin is not None else None
        rc = self._L.oracle_map_load_points(self.h, xyz.ctypes.data, xyz.shape[0], xyz.shape[1],
                                            float(range_limit), _dp(ro) if ro is not None else None)
        if rc:
            raise RuntimeError("oracle_map_load_points")

    def compute_cells(self, n_min=3, eval_factor=1000.0):
        self._L.oracle_map_compute_cells(self.h, int(n_min), float(eval_factor))

    def compute_cells_full(self, n_min=3, eval_factor=1000.0, maxnumpoints=1e5, occupancy_limit=255.0):
        """computeNDTCells(SAMPLE_VARIANCE, maxnumpoints, occupancy_limit, origin, noise) (fuser_hmt.cpp:94, 486)."""
        self._L.oracle_map_compute_cells_full(self.h, int(n_min), float(eval_factor), float(maxnumpoints),
                                              float(occupancy_limit))

    def add_point_cloud(self, origin, xyz, maxz=100.0, sensor_noise=0.1, occupancy_limit=255.0, order_free=False):
        """NDTMap::addPointCloud(origin, cloud, classifierTh, maxz, sensor_noise, occupancy_limit) (fuser_hmt.cpp:92, 485)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        o = _f64(origin)
        rc = self._L.oracle_map_add_point_cloud(self.h, _dp(o), xyz.ctypes.data, xyz.shape[0], xyz.shape[1], float(maxz),
                                                float(sensor_noise), float(occupancy_limit), int(bool(order_free)))
        if rc:
            raise RuntimeError("oracle_map_add_point_cloud")

    def occupancy(self):
        """NDTCell::occ of every slot, shape (sx, sy, sz)."""
        size = (C.c_int * 3)()
        self._L.oracle_map_size(self.h, size)
        out = np.zeros(tuple(size), dtype=np.float32)
        self._L.oracle_map_occupancy(self.h, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def num_cells(self):
        return self._L.oracle_map_num_cells(self.h)

    def export_cells(self):
        n = self.num_cells()
        mean = np.zeros((n, 3))
        cov = np.zeros((n, 3, 3))
        idx = np.zeros((n, 3), dtype=np.int32)
        npts = np.zeros(n, dtype=np.int32)
        self._L.oracle_map_export_cells(self.h, _dp(mean), _dp(cov), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                        npts.ctypes.data_as(C.POINTER(C.c_int32)))
        return mean, cov, idx, npts

    def set_cells(self, mean, cov):
        mean = _f64(mean)
        cov = _f64(cov)
        self._L.oracle_map_set_cells(self.h, _dp(mean), _dp(cov), mean.shape[0])


def derivatives(target, src_mean, src_cov, n_neighbours=2, compute_hessian=True, lfd1=1.0, lfd2=0.05):
    src_mean = _f64(src_mean)
    src_cov = _f64(src_cov)
    g = np.zeros(6)
    H = np.zeros((6, 6))
    s = lib().oracle_derivatives(target.h, _dp(src_mean), _dp(src_cov), src_mean.shape[0], n_neighbours,
                                 int(compute_hessian), lfd1, lfd2, _dp(g), _dp(H))
    return s, g, H


def score_at(target, source, T, p, n_neighbours=2, lfd1=1.0, lfd2=0.05):
    Tc = _f64(np.asarray(T).T.reshape(-1))  # column-major
    p = _f64(p)
    return lib().oracle_score_at(target.h, source.h, _dp(Tc), _dp(p), n_neighbours, lfd1, lfd2)


def match_d2d(target, source, T0, **kw):
    """NDTMatcherD2D::match.  T0: 4x4 numpy (row/col math convention); returns (T, result dict)."""
    prm = dict(DEFAULT_PARAMS)
    prm.update(kw)
    P = MatchParams(**prm)
    R = MatchResult()
    Tc = _f64(np.asarray(T0, dtype=np.float64).T.reshape(-1)).copy()
    rc = lib().oracle_match_d2d(target.h, source.h, _dp(Tc), C.byref(P), C.byref(R))
    if rc:
        raise RuntimeError("oracle_match_d2d rc=%d" % rc)
    T = Tc.reshape(4, 4).T.copy()
    return T, dict(converged=bool(R.converged), iterations=R.iterations, fevals=R.fevals, score=R.score,
                   exit_code=R.exit_code)


def match_fusion(target, source, T0, Tcov, use_soft_constraints=True, tikhonov=False, **kw):
    """ndt_feature::matchFusion with empty feature maps (NDT + odometry soft constraint and / or Tikhonov)."""
    prm = dict(DEFAULT_PARAMS)
    prm.update(kw)
    P = MatchParams(**prm)
    R = MatchResult()
    Tc = _f64(np.asarray(T0, dtype=np.float64).T.reshape(-1)).copy()
    cov = _f64(np.asarray(Tcov, dtype=np.float64).reshape(-1))
    rc = lib().oracle_match_fusion(target.h, source.h, _dp(Tc), C.byref(P), _dp(cov),
                                   int(bool(use_soft_constraints)) | (2 if tikhonov else 0), C.byref(R))
    if rc:
        raise RuntimeError("oracle_match_fusion rc=%d" % rc)
    return Tc.reshape(4, 4).T.copy(), dict(converged=bool(R.converged), iterations=R.iterations, fevals=R.fevals,
                                           score=R.score, exit_code=R.exit_code)


def match_fusion_feat(target, source, T0, Tcov, feat, use_soft_constraints=True, tikhonov=False, step_control_fusion=False, **kw):
    """ndt_feature::matchFusion with feature / odometry-cell maps: feat = (src_mean [k,3], src_cov [k,6], tgt_mean, tgt_cov)."""
    prm = dict(DEFAULT_PARAMS)
    prm.update(kw)
    P = MatchParams(**prm)
    R = MatchResult()
    Tc = _f64(np.asarray(T0, dtype=np.float64).T.reshape(-1)).copy()
    cov = _f64(np.asarray(Tcov, dtype=np.float64).reshape(-1))
    sm, sc, tm, tc = [_f64(np.asarray(a, dtype=np.float64).reshape(-1)) for a in feat]
    k = len(sm) // 3
    L = lib()
    L.oracle_match_fusion_feat.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(MatchParams), C.POINTER(C.c_double),
                                           C.c_int, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                           C.POINTER(C.c_double), C.POINTER(MatchResult)]
    rc = L.oracle_match_fusion_feat(target.h, source.h, _dp(Tc), C.byref(P), _dp(cov),
                                    int(bool(use_soft_constraints)) | (2 if tikhonov else 0) | (4 if step_control_fusion else 0), k, _dp(sm), _dp(sc), _dp(tm), _dp(tc),
                                    C.byref(R))
    if rc:
        raise RuntimeError("oracle_match_fusion_feat rc=%d" % rc)
    return Tc.reshape(4, 4).T.copy(), dict(converged=bool(R.converged), iterations=R.iterations, fevals=R.fevals,
                                           score=R.score, exit_code=R.exit_code)


def tcov_flips(reset=False):
    """How often lineSearchMTFusionTcov's in-place negation of the increment (fusion.h:89-95) has fired (test aid)."""
    L = lib()
    L.oracle_debug_tcov_flips.restype = C.c_long
    L.oracle_debug_tcov_flips.argtypes = [C.c_int]
    return int(L.oracle_debug_tcov_flips(int(bool(reset))))


def set_sum_mode(mode):
    """Order in which derivativesNDT adds the source cells' terms (test knob, see ndt_oracle.c): 0 = the reference's."""
    lib().oracle_set_sum_mode(int(mode))


def pose_to_T(p):
    p = _f64(p)
    T = np.zeros(16)
    lib().oracle_pose_to_T(_dp(p), _dp(T))
    return T.reshape(4, 4).T.copy()


def eig_sym(A):
    A = _f64(A)
    n = A.shape[0]
    ev = np.zeros(n)
    V = np.zeros((n, n))
    lib().oracle_eig_sym(n, _dp(A), _dp(ev), _dp(V))
    return ev, V


def ldlt_solve(A, b):
    A = _f64(A)
    b = _f64(b)
    x = np.zeros_like(b)
    lib().oracle_ldlt_solve(A.shape[0], _dp(A), _dp(b), _dp(x))
    return x


def mahalanobis(x, Q):
    x = _f64(x)
    Q = _f64(Q)
    g = np.zeros(6)
    H = np.zeros((6, 6))
    s = lib().oracle_mahalanobis(_dp(x), _dp(Q), _dp(g), _dp(H))
    return s, g, H


def mt_cstep(stx, fx, dx, sty, fy, dy, stp, fp, dp_, brackt, stmin, stmax):
    v = [C.c_double(z) for z in (stx, fx, dx, sty, fy, dy, stp)]
    b = C.c_int(int(brackt))
    info = lib().oracle_mt_cstep(*[C.byref(z) for z in v], fp, dp_, C.byref(b), stmin, stmax)
    return info, [z.value for z in v], bool(b.value)


def mt_linesearch(phi, finit, dginit):
    """phi(stp) -> (f, dg).  Returns (step, nfev, info)."""
    def _cb(_ctx, stp, dg_out):
        f, dg = phi(stp)
        dg_out[0] = dg
        return f
    cb = PHI_FN(_cb)
    nfev = C.c_int(0)
    info = C.c_int(0)
    stp = lib().oracle_mt_linesearch(cb, None, finit, dginit, C.byref(nfev), C.byref(info))
    return stp, nfev.value, info.value


def overlap_score(ref, mov, T):
    """ndt_feature::overlapNDTOccupancyScore(ref, mov, T) -> (score, nb_sum)."""
    Tc = _f64(np.asarray(T, dtype=np.float64).T.reshape(-1))
    nb = C.c_longlong(0)
    s = lib().oracle_overlap_score(ref.h, mov.h, _dp(Tc), C.byref(nb))
    return s, nb.value


def beam_evidence(mean, cov, origin, end, sensor_noise=0.1):
    """One beam through one Gaussian cell: float log-odds update, or None when the cell is left alone."""
    mean, cov, origin = _f64(mean), _f64(cov), _f64(origin)
    pe = np.ascontiguousarray(end, dtype=np.float32)
    out = C.c_float(0)
    ok = lib().oracle_beam_evidence(_dp(mean), _dp(cov), _dp(origin), pe.ctypes.data_as(C.POINTER(C.c_float)),
                                    float(sensor_noise), C.byref(out))
    return out.value if ok else None


def occupancy_rescaled(occ):
    return float(lib().oracle_occupancy_rescaled(float(occ)))


def covariance(target, source, T, n_neighbours=2, lfd1=1.0, lfd2=0.05, mode=0):
    """NDTMatcherD2D::covariance(target, source, T, cov) -> 6x6."""
    Tc = _f64(np.asarray(T, dtype=np.float64).T.reshape(-1))
    cov = np.zeros((6, 6))
    rc = lib().oracle_covariance(target.h, source.h, _dp(Tc), int(n_neighbours), lfd1, lfd2, int(mode), _dp(cov))
    if rc:
        raise RuntimeError("oracle_covariance rc=%d" % rc)
    return cov
