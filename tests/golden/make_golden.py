#!/usr/bin/env python3
"""Generates tests/golden/golden_v1.npz -- known-answer vectors for the CPU oracle and the HIP path.

The reference holds NO golden vectors for this path ("parity unpinned"), and it
is C++ that cannot be built or imported here, so these vectors come from an INDEPENDENT NumPy
restatement of the published formulas, written without looking at oracle/ndt_oracle.c:

  K6  LazyGrid::getIndexForPoint   idx = floor((p-c)/res + 0.5) + size/2.0 -> int
  K5  NDTCell::computeGaussian / rescaleCovariance   np.cov(ddof=1), np.linalg.eigh, floor at
      lambda_max / 1000
  K2  D2D-NDT score  s = -lfd1 * exp(-lfd2/2 * d^T (R C_i R^T + C_j)^-1 d)  summed over the
      (2n+1)^3 neighbourhood; gradient and Hessian by CENTRAL FINITE DIFFERENCES of that score
      under p -> Trans(p0..2) * Rx(p3) * Ry(p4) * Rz(p5)  (ndt_matcher_d2d_fusion.h:1036-1039)
  K1  Mahalanobis KAT of ndt_feature/src/odom_hessian_test.cpp:272-297
  K7  More-Thuente: MINPACK mcstep (the NOX / perception_oru variant) restated in NumPy,
      cross-checked against scipy.optimize._dcsrch.dcstep on the cases where the 1983 and 1996
      variants coincide (info 2 and 4), and the cvsrch driver with the constants of
      ndt_matcher_d2d_fusion.h:400-408 on 1-D test functions
  K8  force-2D / robust yaw cases of ndt_feature/src/utils_affine_test.cpp:32-58

Run:  python tests/golden/make_golden.py      (NumPy + SciPy only; seconds)
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20170301)


# ---------------------------------------------------------------- K6 index
def index_for_point(p, centre, res, size):
    v = np.floor((np.asarray(p, float) - centre) / res + 0.5) + np.asarray(size) / 2.0
    return np.trunc(v).astype(np.int64)


def gen_index():
    cases = []
    for res, centre, size in [(0.5, (0, 0, 0), (200, 200, 2)), (1.0, (3.2, -1.7, 0.0), (100, 100, 1)),
                              (0.25, (-10.0, 4.0, 1.0), (400, 400, 40)), (1.0, (0, 0, 0), (5, 3, 1))]:
        pts = rng.uniform(-60, 60, size=(40, 3))
        pts[:, 2] = rng.uniform(-1, 1, size=40)
        # boundary points: exactly on cell faces, +- one float32 ulp
        k = rng.integers(-20, 20, size=(12, 3))
        face = np.asarray(centre) + (k + 0.5) * res
        pts = np.concatenate([pts, face, np.nextafter(face.astype(np.float32), np.float32(1e9)).astype(float),
                              np.nextafter(face.astype(np.float32), np.float32(-1e9)).astype(float)])
        pts = pts.astype(np.float32).astype(np.float64)
        idx = np.stack([index_for_point(p, np.asarray(centre, float), res, size) for p in pts])
        cases.append(dict(res=res, centre=np.asarray(centre, float), size=np.asarray(size), pts=pts, idx=idx))
    return cases


# ---------------------------------------------------------------- K5 cell statistics
def cell_gaussian(pts, eval_factor=1000.0):
    pts = np.asarray(pts, np.float64)
    mean = pts.mean(axis=0)
    cov = np.cov(pts.T, ddof=1)
    ev, V = np.linalg.eigh(cov)
    if ev.max() <= 0 or ev.min() <= 1e-9 * ev.max():   # rank deficient: see oracle/ndt_oracle.c rescale_covariance
        return mean, cov, False
    mx = ev.max()
    ev2 = np.where(mx > ev * eval_factor, mx / eval_factor, ev)
    if np.any(ev2 != ev):
        cov = V @ np.diag(ev2) @ V.T
    return mean, cov, True


def gen_cells():
    out = []
    # (a) generic gaussian blob inside one 0.5 m voxel centred at (10.0, -7.5, 0)
    L = np.array([[0.08, 0, 0], [0.03, 0.05, 0], [0.0, 0.0, 0.006]])
    p = (np.array([10.0, -7.5, 0.01]) + rng.normal(size=(200, 3)) @ L.T).astype(np.float32)
    p = p[(np.abs(p[:, 0] - 10.0) < 0.24) & (np.abs(p[:, 1] + 7.5) < 0.24) & (np.abs(p[:, 2]) < 0.24)]
    out.append(p)
    # (b) wall-like: thin in x (noise 0.03), long in y, z jitter 0.02*U  -> eigenvalue floor active
    q = np.stack([3.0 + 0.0005 * rng.normal(size=150), -2.0 + rng.uniform(-0.24, 0.24, size=150),
                  0.02 * rng.uniform(size=150)], axis=1).astype(np.float32)
    out.append(q)
    # (c) four points: smallest full-rank sample
    out.append(np.array([[1.01, 1.02, 0.001], [1.1, 0.95, 0.015], [0.93, 1.07, 0.008], [1.05, 1.1, 0.019]], np.float32))
    # (d) far from the origin (cancellation stress): 45 m
    r = (np.array([45.1, -44.9, 0.0]) + rng.normal(size=(500, 3)) * np.array([0.03, 0.1, 0.005])).astype(np.float32)
    r = r[(np.abs(r[:, 0] - 45.0) < 0.24) & (np.abs(r[:, 1] + 45.0) < 0.24) & (np.abs(r[:, 2]) < 0.24)]
    out.append(r)
    # (e) three points: rank-2 covariance -> no Gaussian;  (f) collinear points -> no Gaussian
    out.append(np.array([[1.01, 1.02, 0.001], [1.1, 0.95, 0.015], [0.93, 1.07, 0.008]], np.float32))
    out.append(np.stack([2.0 + 0.01 * np.arange(8), -1.0 + 0.02 * np.arange(8), 0.01 + 0.001 * np.arange(8)], axis=1).astype(np.float32))
    res = []
    for pts in out:
        mean, cov, ok = cell_gaussian(pts)
        res.append(dict(pts=pts, mean=mean, cov=cov, ok=ok))
    return res


# ---------------------------------------------------------------- K2 D2D score + FD derivatives
def rot(p):
    cx, sx, cy, sy, cz, sz = np.cos(p[3]), np.sin(p[3]), np.cos(p[4]), np.sin(p[4]), np.cos(p[5]), np.sin(p[5])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def d2d_score(p, src_mean, src_cov, tgt_mean, tgt_cov, tgt_idx, grid, nn, lfd1=1.0, lfd2=0.05, fixed_nb=None):
    """Score with source cells moved by TR(p).  Neighbourhood is decided at p (or frozen to
    fixed_nb, the pair list found at p=0, so that finite differences see a smooth function)."""
    res, centre, size = grid
    R = rot(p)
    t = np.asarray(p[:3])
    s = 0.0
    pairs = []
    for i in range(len(src_mean)):
        m = R @ src_mean[i] + t
        C = R @ src_cov[i] @ R.T
        if fixed_nb is None:
            ic = index_for_point(m, centre, res, size)
            js = [j for j in range(len(tgt_mean)) if np.all(np.abs(tgt_idx[j] - ic) <= nn)]
        else:
            js = fixed_nb[i]
        pairs.append(js)
        for j in js:
            d = m - tgt_mean[j]
            B = np.linalg.inv(C + tgt_cov[j])
            s += -lfd1 * np.exp(-lfd2 / 2.0 * d @ B @ d)
    return s, pairs


def gen_d2d():
    res, centre, size = 1.0, np.zeros(3), np.array([40, 40, 4])
    grid = (res, centre, size)
    # target: one gaussian per occupied voxel, means inside their voxel
    occ = rng.choice(12 * 12 * 3, size=60, replace=False)
    tgt_idx = np.stack([occ // 36 + 14, (occ // 3) % 12 + 14, occ % 3 + 1], axis=1)
    tgt_mean = centre + (tgt_idx - size / 2.0) * res + rng.uniform(-0.4, 0.4, size=(60, 3))

    def rand_cov(n, scale):
        A = rng.normal(size=(n, 3, 3)) * scale
        return A @ np.transpose(A, (0, 2, 1)) + 0.01 * np.eye(3)

    tgt_cov = rand_cov(60, 0.25)
    src_mean = centre + rng.uniform(-5.5, 5.5, size=(25, 3)) * np.array([1, 1, 0.2]) + np.array([0.3, -0.2, 0.5])
    src_cov = rand_cov(25, 0.3)
    p0 = np.zeros(6)
    s0, nb = d2d_score(p0, src_mean, src_cov, tgt_mean, tgt_cov, tgt_idx, grid, 2)

    def f(p):
        return d2d_score(p, src_mean, src_cov, tgt_mean, tgt_cov, tgt_idx, grid, 2, fixed_nb=nb)[0]

    h = 1e-5
    g = np.zeros(6)
    for a in range(6):
        e = np.zeros(6); e[a] = h
        g[a] = (f(e) - f(-e)) / (2 * h)
    hh = 1e-4
    H = np.zeros((6, 6))
    for a in range(6):
        for b in range(a, 6):
            ea = np.zeros(6); ea[a] = hh
            eb = np.zeros(6); eb[b] = hh
            H[a, b] = H[b, a] = (f(ea + eb) - f(ea - eb) - f(-ea + eb) + f(-ea - eb)) / (4 * hh * hh)
    n_pairs = sum(len(j) for j in nb)
    return dict(res=res, centre=centre, size=size, tgt_mean=tgt_mean, tgt_cov=tgt_cov, src_mean=src_mean,
                src_cov=src_cov, score=s0, grad_fd=g, hess_fd=H, n_pairs=n_pairs)


# ---------------------------------------------------------------- K7 More-Thuente
def mcstep(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stmin, stmax):
    """MINPACK-1 mcstep (the variant in NOX's MoreThuente and perception_oru)."""
    info = 0
    if (brackt and (stp <= min(stx, sty) or stp >= max(stx, sty))) or dx * (stp - stx) >= 0.0 or stmax < stmin:
        return info, (stx, fx, dx, sty, fy, dy, stp), brackt
    sgnd = dp * (dx / abs(dx))
    if fp > fx:
        info, bound = 1, True
        theta = 3 * (fx - fp) / (stp - stx) + dx + dp
        s = max(abs(theta), abs(dx), abs(dp))
        gamma = s * np.sqrt((theta / s) ** 2 - (dx / s) * (dp / s))
        if stp < stx:
            gamma = -gamma
        p = (gamma - dx) + theta
        q = ((gamma - dx) + gamma) + dp
        r = p / q
        stpc = stx + r * (stp - stx)
        stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2) * (stp - stx)
        stpf = stpc if abs(stpc - stx) < abs(stpq - stx) else stpc + (stpq - stpc) / 2
        brackt = True
    elif sgnd < 0.0:
        info, bound = 2, False
        theta = 3 * (fx - fp) / (stp - stx) + dx + dp
        s = max(abs(theta), abs(dx), abs(dp))
        gamma = s * np.sqrt((theta / s) ** 2 - (dx / s) * (dp / s))
        if stp > stx:
            gamma = -gamma
        p = (gamma - dp) + theta
        q = ((gamma - dp) + gamma) + dx
        r = p / q
        stpc = stp + r * (stx - stp)
        stpq = stp + (dp / (dp - dx)) * (stx - stp)
        stpf = stpc if abs(stpc - stp) > abs(stpq - stp) else stpq
        brackt = True
    elif abs(dp) < abs(dx):
        info, bound = 3, True
        theta = 3 * (fx - fp) / (stp - stx) + dx + dp
        s = max(abs(theta), abs(dx), abs(dp))
        gamma = s * np.sqrt(max(0.0, (theta / s) ** 2 - (dx / s) * (dp / s)))
        if stp > stx:
            gamma = -gamma
        p = (gamma - dp) + theta
        q = (gamma + (dx - dp)) + gamma
        r = p / q
        if r < 0.0 and gamma != 0.0:
            stpc = stp + r * (stx - stp)
        elif stp > stx:
            stpc = stmax
        else:
            stpc = stmin
        stpq = stp + (dp / (dp - dx)) * (stx - stp)
        if brackt:
            stpf = stpc if abs(stp - stpc) < abs(stp - stpq) else stpq
        else:
            stpf = stpc if abs(stp - stpc) > abs(stp - stpq) else stpq
    else:
        info, bound = 4, False
        if brackt:
            theta = 3 * (fp - fy) / (sty - stp) + dy + dp
            s = max(abs(theta), abs(dy), abs(dp))
            gamma = s * np.sqrt((theta / s) ** 2 - (dy / s) * (dp / s))
            if stp > sty:
                gamma = -gamma
            p = (gamma - dp) + theta
            q = ((gamma - dp) + gamma) + dy
            r = p / q
            stpf = stp + r * (sty - stp)
        elif stp > stx:
            stpf = stmax
        else:
            stpf = stmin
    if fp > fx:
        sty, fy, dy = stp, fp, dp
    else:
        if sgnd < 0.0:
            sty, fy, dy = stx, fx, dx
        stx, fx, dx = stp, fp, dp
    stpf = max(stmin, min(stmax, stpf))
    stp = stpf
    if brackt and bound:
        if sty > stx:
            stp = min(stx + 0.66 * (sty - stx), stp)
        else:
            stp = max(stx + 0.66 * (sty - stx), stp)
    return info, (stx, fx, dx, sty, fy, dy, stp), brackt


def cvsrch(phi, finit, dginit):
    """Driver with the constants of ndt_matcher_d2d_fusion.h:400-408; returns (stp, nfev, info)."""
    stp, recovery = 1.0, 0.1
    ftol, gtol, stpmax, stpmin, maxfev, xtol = 0.11111, 0.99999, 4.0, 0.001, 40, 0.01
    info, infoc, brackt, stage1, nfev = 0, 1, False, True, 0
    dgtest = ftol * dginit
    width = stpmax - stpmin
    width1 = 2 * width
    stx, fx, dgx = 0.0, finit, dginit
    sty, fy, dgy = 0.0, finit, dginit
    while True:
        if brackt:
            stmin, stmax = min(stx, sty), max(stx, sty)
        else:
            stmin, stmax = stx, stp + 4 * (stp - stx)
        stp = min(max(stp, stpmin), stpmax)
        if (brackt and (stp <= stmin or stp >= stmax)) or nfev >= maxfev - 1 or infoc == 0 or \
                (brackt and stmax - stmin <= xtol * stmax):
            stp = stx
        f, dg = phi(stp)
        nfev += 1
        ftest1 = finit + stp * dgtest
        if (brackt and (stp <= stmin or stp >= stmax)) or infoc == 0:
            info = 6
        if stp == stpmax and f <= ftest1 and dg <= dgtest:
            info = 5
        if stp == stpmin and (f > ftest1 or dg >= dgtest):
            info = 4
        if nfev >= maxfev:
            info = 3
        if brackt and stmax - stmin <= xtol * stmax:
            info = 2
        if f <= ftest1 and abs(dg) <= gtol * (-dginit):
            info = 1
        if info != 0:
            return (stp if info == 1 else recovery), nfev, info
        if stage1 and f <= ftest1 and dg >= min(ftol, gtol) * dginit:
            stage1 = False
        if stage1 and f <= fx and f > ftest1:
            fm, fxm, fym = f - stp * dgtest, fx - stx * dgtest, fy - sty * dgtest
            dgm, dgxm, dgym = dg - dgtest, dgx - dgtest, dgy - dgtest
            infoc, (stx, fxm, dgxm, sty, fym, dgym, stp), brackt = mcstep(stx, fxm, dgxm, sty, fym, dgym, stp,
                                                                          fm, dgm, brackt, stmin, stmax)
            fx, fy, dgx, dgy = fxm + stx * dgtest, fym + sty * dgtest, dgxm + dgtest, dgym + dgtest
        else:
            infoc, (stx, fx, dgx, sty, fy, dgy, stp), brackt = mcstep(stx, fx, dgx, sty, fy, dgy, stp, f, dg,
                                                                      brackt, stmin, stmax)
        if brackt:
            if abs(sty - stx) >= 0.66 * width1:
                stp = stx + 0.5 * (sty - stx)
            width1 = width
            width = abs(sty - stx)


LS_FUNCS = {
    # name: (phi(stp) -> f, dphi)   all with phi'(0) < 0
    "quad_unit": (lambda t: (t - 1.0) ** 2, lambda t: 2 * (t - 1.0)),
    "quad_far": (lambda t: 0.05 * (t - 9.0) ** 2, lambda t: 0.1 * (t - 9.0)),
    "quad_near": (lambda t: 40.0 * (t - 0.02) ** 2, lambda t: 80.0 * (t - 0.02)),
    "mt1": (lambda t: -t / (t * t + 2.0), lambda t: (t * t - 2.0) / (t * t + 2.0) ** 2),
    "mt2": (lambda t: (t + 0.004) ** 5 - 2 * (t + 0.004) ** 4,
            lambda t: 5 * (t + 0.004) ** 4 - 8 * (t + 0.004) ** 3),
    "expwell": (lambda t: -np.exp(-(t - 2.5) ** 2), lambda t: 2 * (t - 2.5) * np.exp(-(t - 2.5) ** 2)),
    "steep": (lambda t: -np.exp(-200.0 * (t - 0.05) ** 2) - 0.01 * t,
              lambda t: 400.0 * (t - 0.05) * np.exp(-200.0 * (t - 0.05) ** 2) - 0.01),
}


def gen_mt():
    from scipy.optimize._dcsrch import dcstep
    steps = []
    tries = 0
    while len(steps) < 60 and tries < 20000:
        tries += 1
        # consistent data: values/derivatives of a random smooth function at stx < stp < sty
        c = rng.normal(size=5)
        phi = lambda t: c[0] * t + c[1] * t ** 2 + c[2] * t ** 3 + c[3] * np.sin(2 * t + c[4])
        dphi = lambda t: c[0] + 2 * c[1] * t + 3 * c[2] * t ** 2 + 2 * c[3] * np.cos(2 * t + c[4])
        stx = rng.uniform(0, 1)
        stp = stx + rng.uniform(0.05, 2)
        sty = stp + rng.uniform(0.1, 2)
        fx, dx, fy, dy, fp, dp = phi(stx), dphi(stx), phi(sty), dphi(sty), phi(stp), dphi(stp)
        if dx >= -1e-3:
            continue
        brackt = bool(fy > fx or dy > 0)
        stmin, stmax = 0.0, 10.0
        info, new, nb = mcstep(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stmin, stmax)
        if info == 0 or not np.all(np.isfinite(new)):
            continue
        if sum(1 for r in steps if r[12] == info) >= 15:
            continue
        if info in (2, 4):  # 1983 and 1996 variants coincide: cross-check with scipy
            ref = dcstep(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stmin, stmax)
            assert np.allclose(ref[:7], new, rtol=1e-13, atol=1e-13) and bool(ref[7]) == nb, (info, ref, new)
        steps.append(np.array([stx, fx, dx, sty, fy, dy, stp, fp, dp, float(brackt), stmin, stmax, info,
                               *new, float(nb)]))
    ls = {}
    for name, (f, df) in LS_FUNCS.items():
        stp, nfev, info = cvsrch(lambda t: (f(t), df(t)), f(0.0), df(0.0))
        ls[name] = np.array([stp, nfev, info])
    return np.stack(steps), ls


# ---------------------------------------------------------------- K1 / K8
def gen_mahalanobis():
    # odom_hessian_test.cpp:272-297: x=(1,0,2,3,4,5), x0=(0,0,3,4,5,6); f=(x-x0)^T C (x-x0);
    # one full Newton step from x lands on x0 and H*dx = -g.
    A = rng.normal(size=(6, 6))
    Cm = A @ A.T + np.eye(6)
    x = np.array([1.0, 0, 2, 3, 4, 5])
    x0 = np.array([0.0, 0, 3, 4, 5, 6])
    d = x - x0
    H = Cm + Cm.T
    return dict(C=Cm, x=x, x0=x0, score=d @ Cm @ d, grad=H @ d, hess=H)


def main():
    out = {}
    for k, c in enumerate(gen_index()):
        for key, v in c.items():
            out["idx%d_%s" % (k, key)] = np.asarray(v)
    for k, c in enumerate(gen_cells()):
        for key, v in c.items():
            out["cell%d_%s" % (k, key)] = np.asarray(v)
    for key, v in gen_d2d().items():
        out["d2d_" + key] = np.asarray(v)
    steps, ls = gen_mt()
    out["mt_cstep"] = steps
    for name, v in ls.items():
        out["mt_ls_" + name] = v
    for key, v in gen_mahalanobis().items():
        out["maha_" + key] = v
    path = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
