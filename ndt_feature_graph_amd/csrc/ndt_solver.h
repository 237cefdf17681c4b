// ndt_solver.h -- the serial part of NDTMatcherD2D::match as a resumable state machine, compiled for
// BOTH the device (thread 0 of the persistent match kernel) and the host (small batches: the host
// drives one multi-workgroup derivative kernel per evaluation).  Restated from the in-repo copy of the
// Newton loop and More-Thuente driver, ndt_feature/include/ndt_feature/ndt_matcher_d2d_fusion.h:847-1121
// and :390-793 (constants :400-408).
//
// Protocol: init -> { evaluate derivatives at st.Teval (with Hessian iff st.with_h) -> match_state_step }
// until st.done.
#pragma once
#include "ndt_math.h"
#include <float.h>

#ifndef NDT_SOLVER_VGPRS
#define NDT_SOLVER_VGPRS
#endif
#define NDT_HDN static __host__ __device__ __attribute__((noinline)) NDT_SOLVER_VGPRS

enum { PH_NEWTON = 0, PH_LS_TRIAL = 1, PH_FINAL = 2 };

struct MTState {
    double stp, finit, dginit, dgtest, width, width1, stx, fx, dgx, sty, fy, dgy, stmin, stmax;
    int brackt, stage1, nfev, infoc;
};

struct MatchState {
    rigid T, Tbest, Teval;
    double score_best, score_here;
    double incr[6];
    // More-Thuente (fusion.h:400-408, 485-521): one block, copied to registers by the functions that work on it (the
    // state lives in LDS: field-by-field access is one dependent LDS round trip after the other)
    MTState mt;
    int itr_ctr, fevals, ret, exit_code, phase, with_h, done;
    // The first More-Thuente trial (stp = 1) is usually accepted, and the next Newton iteration then evaluates
    // score, gradient AND Hessian at exactly the pose of that trial.  While trials keep being accepted first
    // time (spec_ok) the trial is evaluated with its Hessian and the Newton iteration consumes those sums
    // instead of a second evaluation of the same pose (it still counts as an evaluation in `fevals`).
    int spec_ok, trial_has_h, reuse_sums;
    // matchFusion soft constraint (fusion.h:875-890): X = pose_local_v, Q = Tcov^-1 (row-major)
    int use_prior;
    double pose_local[6];
    double Q[36];
    // matchFusion generalised Tikhonov regularisation (fusion.h:894-911): x0 = 2D pose vector of T Tinit^-1
    int use_tikhonov;
    rigid Tinit_inv;
    double x0[6];
};

// computeScoreMahalanobis (fusion.h:24-27): x^T Q x
NDT_HD double prior_score(const MatchState &st)
{
    double s = 0;
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) s += st.pose_local[i] * st.Q[i * 6 + j] * st.pose_local[j];
    return s;
}

// x0^T Q x0 (fusion.h:910, 1113-1115)
NDT_HD double tikhonov_score(const MatchState &st)
{
    double s = 0;
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) s += st.x0[i] * st.Q[i * 6 + j] * st.x0[j];
    return s;
}

// x0: convertAffineToVector(forceEigenAffine3dTo2d(T Tinit^-1)) (fusion.h:903-907; utils.h:30-68, 161-169) =
// (tx, ty, 0, 0, 0, yaw), yaw by getRobustYawFromAffine3d: acos of the rotated x axis' x component, signed by its y
// component (the argument is clamped to [-1, 1]: one ulp above 1 would be a NaN in the reference)
NDT_HD void tikhonov_x0(MatchState &st)
{
    rigid D;
    rigid_mul(st.T, st.Tinit_inv, D);
    double c = D.r[0];
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    const double ang = acos(c);
    st.x0[0] = D.t[0]; st.x0[1] = D.t[1]; st.x0[2] = 0.0;
    st.x0[3] = 0.0; st.x0[4] = 0.0; st.x0[5] = (D.r[3] > 0) ? ang : -ang;
}

NDT_HD double dmin(double a, double b) { return a < b ? a : b; }
NDT_HD double dmax(double a, double b) { return a > b ? a : b; }
NDT_HD double absmax3(double a, double b, double c) { return dmax(dmax(fabs(a), fabs(b)), fabs(c)); }

// MoreThuente::cstep = MINPACK mcstep (published algorithm: More & Thuente, ACM TOMS 20(3), 1994);
// call sites fusion.h:756,775.
NDT_HD int mt_cstep(double &stx, double &fx, double &dx, double &sty, double &fy, double &dy,
                                     double &stp, double fp, double dp, int &brackt, double stmin, double stmax)
{
    int info = 0;
    bool bound;
    double theta, s, gamma, p, q, r, stpc, stpq, stpf;
    if ((brackt && ((stp <= dmin(stx, sty)) || (stp >= dmax(stx, sty)))) || (dx * (stp - stx) >= 0.0) ||
        (stmax < stmin))
        return info;
    double sgnd = dp * (dx / fabs(dx));
    if (fp > fx) {
        info = 1; bound = true;
        theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
        s = absmax3(theta, dx, dp);
        gamma = s * sqrt(((theta / s) * (theta / s)) - (dx / s) * (dp / s));
        if (stp < stx) gamma = -gamma;
        p = (gamma - dx) + theta;
        q = ((gamma - dx) + gamma) + dp;
        r = p / q;
        stpc = stx + r * (stp - stx);
        stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2) * (stp - stx);
        if (fabs(stpc - stx) < fabs(stpq - stx)) stpf = stpc;
        else stpf = stpc + (stpq - stpc) / 2;
        brackt = 1;
    } else if (sgnd < 0.0) {
        info = 2; bound = false;
        theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
        s = absmax3(theta, dx, dp);
        gamma = s * sqrt(((theta / s) * (theta / s)) - (dx / s) * (dp / s));
        if (stp > stx) gamma = -gamma;
        p = (gamma - dp) + theta;
        q = ((gamma - dp) + gamma) + dx;
        r = p / q;
        stpc = stp + r * (stx - stp);
        stpq = stp + (dp / (dp - dx)) * (stx - stp);
        if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
        else stpf = stpq;
        brackt = 1;
    } else if (fabs(dp) < fabs(dx)) {
        info = 3; bound = true;
        theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
        s = absmax3(theta, dx, dp);
        gamma = s * sqrt(dmax(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
        if (stp > stx) gamma = -gamma;
        p = (gamma - dp) + theta;
        q = (gamma + (dx - dp)) + gamma;
        r = p / q;
        if ((r < 0.0) && (gamma != 0.0)) stpc = stp + r * (stx - stp);
        else if (stp > stx) stpc = stmax;
        else stpc = stmin;
        stpq = stp + (dp / (dp - dx)) * (stx - stp);
        if (brackt) {
            if (fabs(stp - stpc) < fabs(stp - stpq)) stpf = stpc;
            else stpf = stpq;
        } else {
            if (fabs(stp - stpc) > fabs(stp - stpq)) stpf = stpc;
            else stpf = stpq;
        }
    } else {
        info = 4; bound = false;
        if (brackt) {
            theta = 3 * (fp - fy) / (sty - stp) + dy + dp;
            s = absmax3(theta, dy, dp);
            gamma = s * sqrt(((theta / s) * (theta / s)) - (dy / s) * (dp / s));
            if (stp > sty) gamma = -gamma;
            p = (gamma - dp) + theta;
            q = ((gamma - dp) + gamma) + dy;
            r = p / q;
            stpc = stp + r * (sty - stp);
            stpf = stpc;
        } else if (stp > stx)
            stpf = stmax;
        else
            stpf = stmin;
    }
    if (fp > fx) {
        sty = stp; fy = fp; dy = dp;
    } else {
        if (sgnd < 0.0) { sty = stx; fy = fx; dy = dx; }
        stx = stp; fx = fp; dx = dp;
    }
    stpf = dmin(stmax, stpf);
    stpf = dmax(stmin, stpf);
    stp = stpf;
    if (brackt && bound) {
        if (sty > stx) stp = dmin(stx + 0.66 * (sty - stx), stp);
        else stp = dmax(stx + 0.66 * (sty - stx), stp);
    }
    return info;
}

// head of the More-Thuente while(1) body (fusion.h:523-561): pick the trial step and request its evaluation
NDT_HD void mt_request_trial(MatchState &st, MTState &m)
{
    const double stpmax = 4.0, stpmin = 0.001, xtol = 0.01;
    const int maxfev = 40;
    if (m.brackt) {
        m.stmin = dmin(m.stx, m.sty);
        m.stmax = dmax(m.stx, m.sty);
    } else {
        m.stmin = m.stx;
        m.stmax = m.stp + 4 * (m.stp - m.stx);
    }
    m.stp = dmax(m.stp, stpmin);
    m.stp = dmin(m.stp, stpmax);
    if ((m.brackt && ((m.stp <= m.stmin) || (m.stp >= m.stmax))) || (m.nfev >= maxfev - 1) ||
        (m.infoc == 0) || (m.brackt && (m.stmax - m.stmin <= xtol * m.stmax)))
        m.stp = m.stx;
    double pincr[6];
    for (int a = 0; a < 6; a++) pincr[a] = m.stp * st.incr[a];
    rigid ps;
    pose_to_rigid(pincr, ps);
    rigid_mul(ps, st.T, st.Teval);     // trial cells = ps * nextNDT (fusion.h:556-589)
    st.trial_has_h = (m.nfev == 0 && st.spec_ok) ? 1 : 0;
    st.with_h = st.trial_has_h;
    st.phase = PH_LS_TRIAL;
}

// pose update + convergence tests (fusion.h:1032-1080)
NDT_HDN void apply_step(MatchState &st, double step_size, const NdtMatchParamsDev &prm)
{
    double inorm = 0;
    for (int a = 0; a < 6; a++) {
        st.incr[a] *= step_size;
        inorm += st.incr[a] * st.incr[a];
    }
    inorm = sqrt(inorm);
    for (int a = 0; a < 6; a++) st.pose_local[a] += st.incr[a];   // fusion.h:1045
    rigid TR;
    pose_to_rigid(st.incr, TR);
    rigid_mul(TR, st.T, st.T);          // T = TR*T
    bool convergence = false;
    if (st.itr_ctr > 0) convergence = (inorm < prm.delta_score);
    if (st.itr_ctr > prm.itr_max) {
        convergence = true;
        st.ret = 0;
        st.exit_code = 3;
    }
    st.itr_ctr++;
    st.Teval = st.T;
    if (convergence) { st.phase = PH_FINAL; st.with_h = 0; }
    else { st.phase = PH_NEWTON; st.with_h = 1; }
}

NDT_HDN void newton_step(MatchState &st, const double *sums, const NdtMatchParamsDev &prm)
{
    st.fevals++;
    st.score_here = sums[0];
    if (st.use_prior) st.score_here += prior_score(st);   // fusion.h:875-890
    if (!st.use_tikhonov && st.score_here < st.score_best) {   // fusion.h:914-920 (with Tikhonov: after its term, below)
        st.Tbest = st.T;
        st.score_best = st.score_here;
    }
    // Hessian in registers (static indices).  Inactive dofs (NDTMatcherD2D_2D) are decoupled and given
    // the diagonal value of the first active dof, which leaves lambda_min / lambda_max of the active
    // block unchanged (a diagonal entry is a Rayleigh quotient) and yields a zero increment for them.
    double H[6][6], g[6];
    {
        int o = 7;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = a; b < 6; b++) { H[a][b] = sums[o]; H[b][a] = sums[o]; o++; }
    }
    if (st.use_prior) {                    // + computeHessianMahalanobis = Q + Q^T   (fusion.h:11-22)
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = 0; b < 6; b++) H[a][b] += st.Q[a * 6 + b] + st.Q[b * 6 + a];
    }
    double pad = 0.0;
    bool havepad = false;
    double gnorm = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) {
        g[a] = sums[1 + a];
        if (st.use_prior) {                // + computeGradientMahalanobis = (Q + Q^T) X   (fusion.h:29-32)
            double gp = 0;
#pragma unroll
            for (int j = 0; j < 6; j++) gp += (st.Q[a * 6 + j] + st.Q[j * 6 + a]) * st.pose_local[j];
            g[a] += gp;
        }
    }
    if (st.use_tikhonov) {
        // fusion.h:894-911 with P = I:  g <- H^T g + Q x0,  H <- H^T H + Q,  score += x0^T Q x0
        tikhonov_x0(st);
        double g2[6], H2[6][6];
#pragma unroll
        for (int a = 0; a < 6; a++) {
            double s1 = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) s1 += H[k][a] * g[k] + st.Q[a * 6 + k] * st.x0[k];
            g2[a] = s1;
#pragma unroll
            for (int b = 0; b < 6; b++) {
                double s2 = st.Q[a * 6 + b];
#pragma unroll
                for (int k = 0; k < 6; k++) s2 += H[k][a] * H[k][b];
                H2[a][b] = s2;
            }
        }
#pragma unroll
        for (int a = 0; a < 6; a++) {
            g[a] = g2[a];
#pragma unroll
            for (int b = 0; b < 6; b++) H[a][b] = H2[a][b];
        }
        st.score_here += tikhonov_score(st);
        if (st.score_here < st.score_best) {   // fusion.h:914-920
            st.Tbest = st.T;
            st.score_best = st.score_here;
        }
    }
#pragma unroll
    for (int a = 0; a < 6; a++) {
        bool on = (prm.dof_mask >> a) & 1;
        if (!on) g[a] = 0.0;
        if (on && !havepad) { pad = H[a][a]; havepad = true; }
    }
#pragma unroll
    for (int a = 0; a < 6; a++) {
        bool on_a = (prm.dof_mask >> a) & 1;
#pragma unroll
        for (int b = 0; b < 6; b++) {
            bool on_b = (prm.dof_mask >> b) & 1;
            if (!(on_a && on_b)) H[a][b] = (a == b) ? pad : 0.0;
        }
    }
#pragma unroll
    for (int a = 0; a < 6; a++) gnorm += g[a] * g[a];
    gnorm = sqrt(gnorm);
    // fusion.h:922-940.  evals += regularizer with the same regularizer for every eigenvalue, then
    // H = V diag(evals) V^T, i.e. H + regularizer*I: only lambda_min and lambda_max are needed
    // (sym6_extreme_eigs: tridiagonalisation + Laguerre, no eigenvectors).  A positive definite H (the
    // usual case near the optimum) is certified by an unpivoted Cholesky and skips even that.
    double Lf[6][6], Ldinv[6];
    const bool is_pd = chol_is_pd<6>(H, Lf, Ldinv);
    if (!is_pd) {
        double minC, maxC;
        sym6_extreme_eigs(H, minC, maxC);
        if (minC < 0) {
            double regularizer = gnorm;
            regularizer = (regularizer + minC > 0) ? regularizer : 0.001 * maxC - minC;
#pragma unroll
            for (int a = 0; a < 6; a++) H[a][a] += regularizer;
        }
    }
    if (gnorm <= prm.delta_score) {        // fusion.h:943-965
        if (st.score_here > st.score_best) st.T = st.Tbest;
        st.exit_code = 1;
        st.done = 1;
        return;
    }
    // fusion.h:966  pose_increment_v = -Hessian.ldlt().solve(score_gradient).  The padded 6x6 system
    // performs exactly the arithmetic of the active block (the padding is decoupled, its solution 0).
    // A Hessian that the Cholesky test certified positive definite is solved with that factor (same
    // solution up to rounding, a fraction of the pivoted factorisation's serial latency).
    double dxs[6];
    if (is_pd) chol_solve<6>(Lf, Ldinv, g, dxs);
    else ldlt_solve_static6(H, g, dxs);
    double dginit = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) {
        bool on = (prm.dof_mask >> a) & 1;
        double d = on ? -dxs[a] : 0.0;
        st.incr[a] = d;
        dginit += d * g[a];
    }
    if (dginit > 0) {                      // fusion.h:976-997
        if (st.score_here > st.score_best) st.T = st.Tbest;
        st.exit_code = 2;
        st.done = 1;
        return;
    }
    if (!prm.step_control) {
        apply_step(st, 1.0, prm);
        return;
    }
    // lineSearchMT: its initial derivativesNDT(nextNDT) equals this evaluation (same cells), so the
    // score and gradient are reused instead of being recomputed (fusion.h:444-453).
    // With the soft constraint the reference also runs lineSearchMTFusionTcov first (fusion.h:1008-1010) but
    // throws its step away (:1018-1023: step_size = max(step_size_ndt, 0)); its only possible side effect,
    // flipping the increment when dginit >= 0, cannot trigger here because dginit <= 0 was just checked on
    // the same total gradient.  The step is decided by the NDT-only line search on the NDT-only score.
    MTState m;
    m.finit = sums[0];
    m.dginit = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) m.dginit += st.incr[a] * sums[1 + a];
    if (!st.use_prior && !st.use_tikhonov) m.dginit = dginit;
    if (m.dginit >= 0.0) {                // fusion.h:456-479
        for (int a = 0; a < 6; a++) st.incr[a] = -st.incr[a];
        m.dginit = -m.dginit;
        if (m.dginit >= 0.0) {
            apply_step(st, 0.1, prm);
            return;
        }
    }
    m.stp = 1.0;
    m.brackt = 0; m.stage1 = 1; m.nfev = 0; m.infoc = 1;
    m.dgtest = 0.11111 * m.dginit;
    m.width = 4.0 - 0.001;
    m.width1 = 2 * m.width;
    m.stx = 0.0; m.fx = m.finit; m.dgx = m.dginit;
    m.sty = 0.0; m.fy = m.finit; m.dgy = m.dginit;
    mt_request_trial(st, m);
    st.mt = m;
}

// tail of the More-Thuente while(1) body after the trial evaluation (fusion.h:637-790)
// last evaluation at the returned pose (fusion.h:1085-1121)
NDT_HD void match_state_final(MatchState &st, const double *sums)
{
    st.fevals++;
    st.score_here = sums[0];
    if (st.use_prior) st.score_here += prior_score(st);   // fusion.h:1098-1110
    if (st.use_tikhonov) st.score_here += tikhonov_score(st);   // fusion.h:1113-1115: x0 of the LAST Newton evaluation
    if (st.score_here > st.score_best) st.T = st.Tbest;
    st.done = 1;
}

NDT_HDN void linesearch_step(MatchState &st, const double *sums, const NdtMatchParamsDev &prm)
{
    const double ftol = 0.11111, gtol = 0.99999, stpmax = 4.0, stpmin = 0.001, xtol = 0.01, recoverystep = 0.1;
    const int maxfev = 40;
    MTState m = st.mt;
    st.fevals++;
    double f = sums[0];
    double dg = 0;
    for (int a = 0; a < 6; a++) dg += st.incr[a] * sums[1 + a];
    m.nfev++;
    double ftest1 = m.finit + m.stp * m.dgtest;
    int info = 0;
    if ((m.brackt && ((m.stp <= m.stmin) || (m.stp >= m.stmax))) || (m.infoc == 0)) info = 6;
    if ((m.stp == stpmax) && (f <= ftest1) && (dg <= m.dgtest)) info = 5;
    if ((m.stp == stpmin) && ((f > ftest1) || (dg >= m.dgtest))) info = 4;
    if (m.nfev >= maxfev) info = 3;
    if (m.brackt && (m.stmax - m.stmin <= xtol * m.stmax)) info = 2;
    if ((f <= ftest1) && (fabs(dg) <= gtol * (-m.dginit))) info = 1;
    if (info != 0) {
        const bool first_accepted = (info == 1) && (m.nfev == 1);

        const bool reuse = first_accepted && st.trial_has_h;      // sums hold the Hessian at the accepted pose
        st.spec_ok = first_accepted ? 1 : 0;
        st.mt = m;
        apply_step(st, (info == 1) ? m.stp : recoverystep, prm);
        st.reuse_sums = (reuse && !st.done) ? 1 : 0;   // match_state_step consumes the sums once more
        return;
    }
    if (m.stage1 && (f <= ftest1) && (dg >= dmin(ftol, gtol) * m.dginit)) m.stage1 = 0;
    if (m.stage1 && (f <= m.fx) && (f > ftest1)) {
        double fm = f - m.stp * m.dgtest;
        double fxm = m.fx - m.stx * m.dgtest;
        double fym = m.fy - m.sty * m.dgtest;
        double dgm = dg - m.dgtest;
        double dgxm = m.dgx - m.dgtest;
        double dgym = m.dgy - m.dgtest;
        m.infoc = mt_cstep(m.stx, fxm, dgxm, m.sty, fym, dgym, m.stp, fm, dgm, m.brackt, m.stmin, m.stmax);
        m.fx = fxm + m.stx * m.dgtest;
        m.fy = fym + m.sty * m.dgtest;
        m.dgx = dgxm + m.dgtest;
        m.dgy = dgym + m.dgtest;
    } else {
        m.infoc = mt_cstep(m.stx, m.fx, m.dgx, m.sty, m.fy, m.dgy, m.stp, f, dg, m.brackt, m.stmin, m.stmax);
    }
    if (m.brackt) {
        if (fabs(m.sty - m.stx) >= 0.66 * m.width1) m.stp = m.stx + 0.5 * (m.sty - m.stx);
        m.width1 = m.width;
        m.width = fabs(m.sty - m.stx);
    }
    mt_request_trial(st, m);
    st.mt = m;
}


// T0: column-major 4x4 (Eigen::Affine3d storage) or NULL for identity
// Q36: Tcov^-1 row-major, or NULL for NDTMatcherD2D::match; prm.fusion_flags says what it is used for
NDT_HD void match_state_init(MatchState &st, const double *T16, const NdtMatchParamsDev &prm, const double *Q36 = nullptr)
{
    st.use_prior = Q36 != nullptr && (prm.fusion_flags & 1);
    st.use_tikhonov = Q36 != nullptr && (prm.fusion_flags & 2);
    for (int a = 0; a < 6; a++) st.x0[a] = 0.0;
    for (int a = 0; a < 6; a++) st.pose_local[a] = 0.0;
    for (int a = 0; a < 36; a++) st.Q[a] = Q36 ? Q36[a] : 0.0;
    rigid T0;
    if (prm.use_initial_guess && T16) {
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) T0.r[r * 3 + c] = T16[c * 4 + r];
            T0.t[r] = T16[12 + r];
        }
    } else {
        for (int k = 0; k < 9; k++) T0.r[k] = (k % 4 == 0) ? 1.0 : 0.0;
        T0.t[0] = T0.t[1] = T0.t[2] = 0.0;
    }
    st.T = T0; st.Tbest = T0; st.Teval = T0;
    for (int r = 0; r < 3; r++) {          // Tinit^-1 (rigid)
        for (int c = 0; c < 3; c++) st.Tinit_inv.r[r * 3 + c] = T0.r[c * 3 + r];
        st.Tinit_inv.t[r] = -(T0.r[0 * 3 + r] * T0.t[0] + T0.r[1 * 3 + r] * T0.t[1] + T0.r[2 * 3 + r] * T0.t[2]);
    }
    st.score_best = DBL_MAX; st.score_here = 0;
    st.itr_ctr = 0; st.fevals = 0; st.ret = 1; st.exit_code = 0;
    st.phase = PH_NEWTON; st.with_h = 1; st.done = 0;
    st.spec_ok = 0; st.trial_has_h = 0; st.reuse_sums = 0;   // the first line search is rarely a full step
    if ((prm.dof_mask & 0x3f) == 0 || prm.n_neighbours < 0 || prm.n_neighbours > 3) { st.done = 1; st.ret = 0; st.exit_code = -1; }
}

// consumes the sums of the evaluation that was requested (sums[0] score, [1..6] gradient, [7..27] upper
// triangle of the Hessian) and either requests the next evaluation or finishes
NDT_HD void match_state_step(MatchState &st, const double *sums, const NdtMatchParamsDev &prm)
{
    if (st.phase == PH_LS_TRIAL) {
        st.reuse_sums = 0;
        linesearch_step(st, sums, prm);
        // an accepted first trial that was evaluated with its Hessian: the evaluation apply_step just requested
        // is the one these sums come from (same cells, same pose)
        if (!st.reuse_sums) return;
    }
    if (st.phase == PH_NEWTON) newton_step(st, sums, prm);
    else if (st.phase == PH_FINAL) match_state_final(st, sums);
}

NDT_HD void match_state_result(const MatchState &st, double *T16, NdtMatchResultDev &o)
{
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T16[c * 4 + r] = st.T.r[r * 3 + c];
        T16[12 + r] = st.T.t[r];
        T16[r * 4 + 3] = 0.0;
    }
    T16[15] = 1.0;
    o.converged = st.ret;
    o.iterations = st.itr_ctr;
    o.fevals = st.fevals;
    o.exit_code = st.exit_code;
    o.score = (st.score_here > st.score_best) ? st.score_best : st.score_here;
}
