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
#ifdef NDT_SOLVER_INLINE
#define NDT_HDN NDT_HD
#else
#define NDT_HDN static __host__ __device__ __attribute__((noinline)) NDT_SOLVER_VGPRS
#endif

enum { PH_NEWTON = 0, PH_LS_TRIAL = 1, PH_FINAL = 2 };
// what a stage of the solver asks its caller to run next (stages never call each other: a value that lives across a call
// sits in a callee-saved register, and saving those is the only thing that would give the kernels a stack)
enum { NEXT_NONE = 0, NEXT_APPLY_STEP = 1, NEXT_REQUEST_TRIAL = 2 };

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
    double step_size;      // NEXT_APPLY_STEP: the step apply_step takes
    int itr_ctr, fevals, ret, exit_code, phase, with_h, done;
    // The first More-Thuente trial (stp = 1) is usually accepted, and the next Newton iteration then evaluates
    // score, gradient AND Hessian at exactly the pose of that trial.  While trials keep being accepted first
    // time (spec_ok) the trial is evaluated with its Hessian and the Newton iteration consumes those sums
    // instead of a second evaluation of the same pose (it still counts as an evaluation in `fevals`).
    int spec_ok, trial_has_h, reuse_sums;
    // ... unless accepting that trial would END the registration (apply_step: |increment| < DELTA_SCORE after the first
    // iteration, or the iteration cap): nobody reads a Hessian then.  That was one wasted Hessian evaluation per converged
    // registration -- 2.5 of the 10.6 evaluations with a Hessian of a bench registration were never consumed.
    int spec_itr_max;
    // The line search ended on a trial that met the More-Thuente conditions (info 1): the pose apply_step moves to IS that trial's
    // pose (the same products, the same rigid product), so when the registration ends there the score the reference evaluates
    // once more at the returned pose (fusion.h:1085) is the score of the trial just summed -- bit for bit -- and the final
    // evaluation is not run (it still counts in `fevals`).
    int final_from_trial;
    double spec_delta;
    // matchFusion soft constraint (fusion.h:875-890): X = pose_local_v, Q = Tcov^-1 (row-major)
    int use_prior;
    double pose_local[6];
    double Q[36];
    // matchFusion feature / odometry-cell terms (fusion.h:858-871, 1013-1023): the feature maps' sums are added to the
    // NDT sums in the Newton system; the step is the smaller of the two line searches (NDT, then features)
    int use_feat;
    // lineSearchMTFusion (fusion.h:390-793: useFeat, step_control_fusion, no soft constraint): ONE search on
    // f_ndt(trial) + f_feat, where the feature maps are evaluated on the UN-stepped cells (fusion.h:619 passes sourceNDT_feat,
    // not sourceNDTHere_feat): their score and gradient at the current pose are constants of the search
    int ls_joint;
    double ls_fconst, ls_gconst[6];
    double step_ndt;       // result of the NDT line search while the feature line search runs
    int fevals_saved, pad_feat;
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

// head of the More-Thuente while(1) body (fusion.h:523-561): pick the trial step and request its evaluation.
// A stage of its own (the More-Thuente block is handed over in st.mt): the trial pose costs three rigid transforms
// in registers.
// EVERY stage reads what it needs FIRST, into locals, and stores at its end: the state lives in LDS behind a generic pointer,
// a value that is only loaded where the code needs it costs a full round trip (the flat loads of a stage are waited for
// all together), and a stage that loaded field by field between its branches and stores spent most of its time in those
// waits (mt_request_trial: 16 of them, 3.3 k clocks for ~150 executed instructions; round 5).
NDT_HDN void mt_request_trial(MatchState &st)
{
    const double stpmax = 4.0, stpmin = 0.001, xtol = 0.01;
    const int maxfev = 40;
    double stp = st.mt.stp, stmin, stmax;
    const double stx = st.mt.stx, sty = st.mt.sty;
    const int brackt = st.mt.brackt, nfev = st.mt.nfev, infoc = st.mt.infoc;
    double incr[6];
    for (int a = 0; a < 6; a++) incr[a] = st.incr[a];
    const rigid T = st.T;
    const int spec_ok = st.spec_ok, use_feat = st.use_feat, itr_ctr = st.itr_ctr, spec_itr_max = st.spec_itr_max;
    const double spec_delta = st.spec_delta;
    if (brackt) {
        stmin = dmin(stx, sty);
        stmax = dmax(stx, sty);
    } else {
        stmin = stx;
        stmax = stp + 4 * (stp - stx);
    }
    stp = dmax(stp, stpmin);
    stp = dmin(stp, stpmax);
    if ((brackt && ((stp <= stmin) || (stp >= stmax))) || (nfev >= maxfev - 1) ||
        (infoc == 0) || (brackt && (stmax - stmin <= xtol * stmax)))
        stp = stx;
    double pincr[6];
    for (int a = 0; a < 6; a++) pincr[a] = stp * incr[a];
    rigid ps, Te;
    pose_to_rigid(pincr, ps);
    rigid_mul(ps, T, Te);              // trial cells = ps * nextNDT (fusion.h:556-589)
    // Which trials are evaluated WITH their Hessian: the first one while first trials keep being accepted (spec_ok), and every
    // trial from the SECOND on (tools/spec_stats.py, 128 bench pairs: trials per search 1 / 2 / 3 / 4+: 434 / 454 / 244 / 24; the
    // second trial ends 63 % of the searches that reach it, the third 91 %).  An evaluation with the Hessian that is consumed
    // saves a whole gradient-only one; one that is not costs the difference -- worth it above 47 % now that a share task
    // with the Hessian is 1.9 times a gradient-only one (it was 2.4 times, break-even 55 %, before the round's leaner Hessian
    // term: from the third trial on then).  Measured, 100 steps of the bench: from the second trial 646 k registrations/s,
    // from the third 620 k.  NDT_SPEC_FROM: the number of trials before the first speculated one (experiments).
#ifndef NDT_SPEC_FROM
#define NDT_SPEC_FROM 1
#endif
    int spec = ((nfev == 0 && spec_ok) || (!use_feat && nfev >= NDT_SPEC_FROM)) ? 1 : 0;
    if (spec && !use_feat) {
        // accepted, apply_step takes the step stp * incr and ends the registration when that is shorter than DELTA_SCORE (after
        // the first iteration) or the iteration cap is reached: nobody reads a Hessian then
        double n2 = 0;
        for (int a = 0; a < 6; a++) n2 += pincr[a] * pincr[a];
        if (itr_ctr > spec_itr_max || (itr_ctr > 0 && sqrt(n2) < spec_delta)) spec = 0;
    }
    st.mt.stp = stp; st.mt.stmin = stmin; st.mt.stmax = stmax;
    st.Teval = Te;
    st.trial_has_h = spec;
    st.with_h = spec;
    st.phase = PH_LS_TRIAL;
}

// pose update + convergence tests (fusion.h:1032-1080)
NDT_HDN void apply_step(MatchState &st, const NdtMatchParamsDev &prm)
{
    const double step_size = st.step_size;
    double incr[6], pl[6];
    for (int a = 0; a < 6; a++) { incr[a] = st.incr[a]; pl[a] = st.pose_local[a]; }
    const rigid T = st.T;
    const int itr_ctr = st.itr_ctr, itr_max = prm.itr_max;
    const double delta_score = prm.delta_score;
    double inorm = 0;
    for (int a = 0; a < 6; a++) {
        incr[a] *= step_size;
        inorm += incr[a] * incr[a];
    }
    inorm = sqrt(inorm);
    for (int a = 0; a < 6; a++) pl[a] += incr[a];   // fusion.h:1045
    rigid TR, Tn;
    pose_to_rigid(incr, TR);
    rigid_mul(TR, T, Tn);               // T = TR*T
    bool convergence = false;
    if (itr_ctr > 0) convergence = (inorm < delta_score);
    const bool capped = itr_ctr > itr_max;
    if (capped) convergence = true;
    for (int a = 0; a < 6; a++) { st.incr[a] = incr[a]; st.pose_local[a] = pl[a]; }
    st.T = Tn;
    st.Teval = Tn;
    if (capped) { st.ret = 0; st.exit_code = 3; }
    st.itr_ctr = itr_ctr + 1;
    st.phase = convergence ? PH_FINAL : PH_NEWTON;
    st.with_h = convergence ? 0 : 1;
}

// ---- one Newton iteration (fusion.h:857-1031) in STAGES ------------------------------------------------------
// The stages are separate functions (not inlined on the device) that hand their results over through a NewtonWs
// block (LDS on the device): each stage then keeps its own 6x6 matrices in registers.  As one function the iteration
// needs more than the 256 VGPRs of a wave and spills to scratch memory; as stages no stage leaves the caller-saved
// registers, so the kernels that run the solver have no private segment at all.
struct NewtonWs {
    double H[36];           // the system matrix (row-major; after soft constraint / Tikhonov / dof padding)
    double g[6];            // the gradient the system is solved for (scg, fusion.h:913)
    double gpre[6];         // NDT + soft-constraint gradient before the Tikhonov transformation: what
                            // lineSearchMTFusionTcov evaluates at the current pose (fusion.h:82-89)
    double dx[6];           // H^-1 g
    double H2[36];          // H^T H + Q of the Tikhonov stage
    double gnorm;
    int is_pd, pad;
};

// inactive dofs (NDTMatcherD2D_2D): decoupled and given the diagonal value of the first active dof, which leaves
// lambda_min / lambda_max of the active block unchanged (a diagonal entry is a Rayleigh quotient) and yields a zero
// increment for them; then the system goes to the workspace
NDT_HD void newton_mask(double (&H)[6][6], double (&g)[6], const NdtMatchParamsDev &prm, NewtonWs &ws)
{
    double pad = 0.0;
    bool havepad = false;
#pragma unroll
    for (int a = 0; a < 6; a++) {
        bool on = (prm.dof_mask >> a) & 1;
        if (!on) g[a] = 0.0;
        if (on && !havepad) { pad = H[a][a]; havepad = true; }
    }
    double gnorm = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) {
        bool on_a = (prm.dof_mask >> a) & 1;
#pragma unroll
        for (int b = 0; b < 6; b++) {
            bool on_b = (prm.dof_mask >> b) & 1;
            if (!(on_a && on_b)) H[a][b] = (a == b) ? pad : 0.0;
            ws.H[a * 6 + b] = H[a][b];
        }
        ws.g[a] = g[a];
        gnorm += g[a] * g[a];
    }
    ws.gnorm = sqrt(gnorm);
}

// stage 1: score, best pose, Hessian / gradient assembly (fusion.h:857-920), inactive dofs, gradient norm
NDT_HDN void newton_assemble(MatchState &st, const double *sums, const double *fsums, const NdtMatchParamsDev &prm, NewtonWs &ws)
{
    // (inputs first, see mt_request_trial: the 28 sums, the flags, the best score and the pose that may become the best one)
    double s[28];
#pragma unroll
    for (int k = 0; k < 28; k++) s[k] = sums[k];
    const int use_feat = st.use_feat, use_prior = st.use_prior, use_tikhonov = st.use_tikhonov, fevals = st.fevals;
    const double score_best = st.score_best;
    const rigid Tcur = st.T;
    if (use_feat) {                                       // fusion.h:863-871: score, gradient and Hessian += those of the feature maps
#pragma unroll
        for (int k = 0; k < 28; k++) s[k] += fsums[k];
    }
    double score_here = s[0];
    if (use_prior) score_here += prior_score(st);         // fusion.h:875-890
    st.fevals = fevals + 1;
    st.score_here = score_here;
    if (!use_tikhonov && score_here < score_best) {       // fusion.h:914-920 (with Tikhonov: after its term, below)
        st.Tbest = Tcur;
        st.score_best = score_here;
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
            for (int b = a; b < 6; b++) {
                const double h = s[o];
                H[a][b] = h; H[b][a] = h; o++;
            }
    }
    if (use_prior) {                       // + computeHessianMahalanobis = Q + Q^T   (fusion.h:11-22)
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = 0; b < 6; b++) H[a][b] += st.Q[a * 6 + b] + st.Q[b * 6 + a];
    }
#pragma unroll
    for (int a = 0; a < 6; a++) {
        g[a] = s[1 + a];
        if (use_prior) {                   // + computeGradientMahalanobis = (Q + Q^T) X   (fusion.h:29-32)
            double gp = 0;
#pragma unroll
            for (int j = 0; j < 6; j++) gp += (st.Q[a * 6 + j] + st.Q[j * 6 + a]) * st.pose_local[j];
            g[a] += gp;
        }
        ws.gpre[a] = g[a];
    }
    if (use_tikhonov) {                    // (stage 1b takes over: the two matrices of H^T H do not fit beside this one)
#pragma unroll
        for (int a = 0; a < 6; a++) {
            ws.g[a] = g[a];
#pragma unroll
            for (int b = 0; b < 6; b++) ws.H[a * 6 + b] = H[a][b];
        }
        return;
    }
    newton_mask(H, g, prm, ws);
}

// stage 1b: the generalised Tikhonov regularisation (fusion.h:894-911 with P = I):  g <- H^T g + Q x0,
// H <- H^T H + Q,  score += x0^T Q x0.  Matrices stay in the workspace (loops, not registers: matchFusion only).
NDT_HDN void newton_tikhonov(MatchState &st, const NdtMatchParamsDev &prm, NewtonWs &ws)
{
    tikhonov_x0(st);
#pragma unroll 1
    for (int a = 0; a < 6; a++) {
        double s1 = 0;
#pragma unroll 1
        for (int k = 0; k < 6; k++) s1 += ws.H[k * 6 + a] * ws.g[k] + st.Q[a * 6 + k] * st.x0[k];
        ws.dx[a] = s1;                      // (dx is free until stage 2)
#pragma unroll 1
        for (int b = 0; b < 6; b++) {
            double s2 = st.Q[a * 6 + b];
#pragma unroll 1
            for (int k = 0; k < 6; k++) s2 += ws.H[k * 6 + a] * ws.H[k * 6 + b];
            ws.H2[a * 6 + b] = s2;
        }
    }
    st.score_here += tikhonov_score(st);
    if (st.score_here < st.score_best) {   // fusion.h:914-920
        st.Tbest = st.T;
        st.score_best = st.score_here;
    }
    double H[6][6], g[6];
#pragma unroll
    for (int a = 0; a < 6; a++) {
        g[a] = ws.dx[a];
#pragma unroll
        for (int b = 0; b < 6; b++) H[a][b] = ws.H2[a * 6 + b];
    }
    newton_mask(H, g, prm, ws);
}

// stage 2: is H positive definite (the usual case near the optimum)?  An unpivoted Cholesky certifies it, and its
// factor then solves the Newton system (same solution as the pivoted LDL^T up to rounding, a fraction of its serial
// latency); otherwise stages 3 and 4 run.
NDT_HDN void newton_factor(NewtonWs &ws)
{
    double H[6][6], Lf[6][6], Ldinv[6], g[6], dx[6];
#pragma unroll
    for (int a = 0; a < 6; a++) {
        g[a] = ws.g[a];
#pragma unroll
        for (int b = 0; b < 6; b++) H[a][b] = ws.H[a * 6 + b];
    }
    const bool is_pd = chol_is_pd<6>(H, Lf, Ldinv);
    ws.is_pd = is_pd ? 1 : 0;
    if (is_pd) {
        chol_solve<6>(Lf, Ldinv, g, dx);
#pragma unroll
        for (int a = 0; a < 6; a++) ws.dx[a] = dx[a];
    }
}

// stage 3 (H not positive definite): fusion.h:922-940.  evals += regularizer with the same regularizer for every
// eigenvalue, then H = V diag(evals) V^T, i.e. H + regularizer * I: only lambda_min and lambda_max are needed
// (sym6_extreme_eigs: tridiagonalisation + Laguerre, no eigenvectors).
NDT_HDN void newton_regularize(NewtonWs &ws)
{
    double H[6][6];
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = 0; b < 6; b++) H[a][b] = ws.H[a * 6 + b];
    double minC, maxC;
    sym6_extreme_eigs(H, minC, maxC);
    if (minC < 0) {
        double regularizer = ws.gnorm;
        regularizer = (regularizer + minC > 0) ? regularizer : 0.001 * maxC - minC;
#pragma unroll
        for (int a = 0; a < 6; a++) ws.H[a * 7] = H[a][a] + regularizer;
    }
}

// stage 4 (H not positive definite): fusion.h:966  Hessian.ldlt().solve(score_gradient).  The padded 6x6 system
// performs exactly the arithmetic of the active block (the padding is decoupled, its solution 0).
NDT_HDN void newton_ldlt(NewtonWs &ws)
{
    double a[21], y[6];
    ndt_static_for<6>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        y[i] = ws.g[i];
        ndt_static_for<6>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            if constexpr (j <= i) a[ndt_ldlt6::tri(i, j)] = 0.5 * (ws.H[i * 6 + j] + ws.H[j * 6 + i]);
        });
    });
    ndt_ldlt6::solve_packed(a, y);
    ndt_static_for<6>([&](auto I) __attribute__((always_inline)) { ws.dx[decltype(I)::value] = y[decltype(I)::value]; });
}

// head of lineSearchMT (fusion.h:444-521) on the function whose value and gradient at the current pose are finit / g6:
// direction test (the increment is negated IN PLACE when it points uphill, recovery step when it still does), then the
// More-Thuente block.  NEXT_APPLY_STEP: the search is over (st.step_size = recovery step); NEXT_REQUEST_TRIAL: st.mt is set up.
// start of a More-Thuente search along `incr` (in registers; the caller stores it): fusion.h:444-521
NDT_HD int mt_start_local(MatchState &st, double finit, const double *g6, double (&incr)[6])
{
    MTState m;
    m.finit = finit;
    m.dginit = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) m.dginit += incr[a] * g6[a];
    if (m.dginit >= 0.0) {                // fusion.h:456-479
#pragma unroll
        for (int a = 0; a < 6; a++) incr[a] = -incr[a];
        m.dginit = -m.dginit;
        if (m.dginit >= 0.0) {
            st.step_size = 0.1;
            return NEXT_APPLY_STEP;
        }
    }
    m.stp = 1.0;
    m.brackt = 0; m.stage1 = 1; m.nfev = 0; m.infoc = 1;
    m.dgtest = 0.11111 * m.dginit;
    m.width = 4.0 - 0.001;
    m.width1 = 2 * m.width;
    m.stx = 0.0; m.fx = m.finit; m.dgx = m.dginit;
    m.sty = 0.0; m.fy = m.finit; m.dgy = m.dginit;
    m.stmin = 0.0; m.stmax = 0.0;         // (set by mt_request_trial before anybody reads them)
    st.mt = m;
    return NEXT_REQUEST_TRIAL;
}
NDT_HD int mt_start(MatchState &st, double finit, const double *g6)
{
    double incr[6];
    for (int a = 0; a < 6; a++) incr[a] = st.incr[a];
    const int next = mt_start_local(st, finit, g6, incr);
    for (int a = 0; a < 6; a++) st.incr[a] = incr[a];
    return next;
}

// stage 5: the increment, the direction tests and the start of the line search (fusion.h:966-1031, 444-521)
NDT_HDN int newton_finish(MatchState &st, const double *sums, const NdtMatchParamsDev &prm, const NewtonWs &ws)
{
    // (inputs first, see mt_request_trial)
    double dx[6], wg[6], gpre[6], gls[6];
#pragma unroll
    for (int a = 0; a < 6; a++) {
        dx[a] = ws.dx[a]; wg[a] = ws.g[a]; gpre[a] = ws.gpre[a];
        gls[a] = sums[1 + a] + st.ls_gconst[a];            // (+ 0 unless lineSearchMTFusion)
    }
    const double f0 = sums[0] + st.ls_fconst;
    const int dof_mask = prm.dof_mask, step_control = prm.step_control, use_prior = st.use_prior;
    double incr[6];
    double dginit = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) {
        bool on = (dof_mask >> a) & 1;
        double d = on ? -dx[a] : 0.0;
        incr[a] = d;
        dginit += d * wg[a];
    }
    if (dginit > 0) {                      // fusion.h:976-997
#pragma unroll
        for (int a = 0; a < 6; a++) st.incr[a] = incr[a];
        if (st.score_here > st.score_best) st.T = st.Tbest;
        st.exit_code = 2;
        st.done = 1;
        return NEXT_NONE;
    }
    if (!step_control) {
#pragma unroll
        for (int a = 0; a < 6; a++) st.incr[a] = incr[a];
        st.step_size = 1.0;
        return NEXT_APPLY_STEP;
    }
    // With the soft constraint the reference first runs lineSearchMTFusionTcov (fusion.h:1008-1010) and throws its
    // step away (:1018-1023: step_size = max(step_size_ndt, 0)).  What survives is its side effect on the increment,
    // which it takes by reference: when increment . (g_ndt + g_mahalanobis) >= 0 the increment is NEGATED IN PLACE
    // (fusion.h:89-95) before lineSearchMT sees it.  Without Tikhonov that gradient is the one dginit <= 0 was just
    // tested on (only dginit == 0 flips); with Tikhonov scg = H^T g + Q x0 is another vector and the flip is live.
    if (use_prior) {
        double dtcov = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) dtcov += incr[a] * gpre[a];
        if (dtcov >= 0.0) {
#pragma unroll
            for (int a = 0; a < 6; a++) incr[a] = -incr[a];
        }
    }
    // lineSearchMT: its initial derivativesNDT(nextNDT) equals this evaluation (same cells), so the
    // score and gradient are reused instead of being recomputed (fusion.h:444-453).  The step is decided by the
    // NDT-only line search on the NDT-only score.
    const int next = mt_start_local(st, f0, gls, incr);
#pragma unroll
    for (int a = 0; a < 6; a++) st.incr[a] = incr[a];
    return next;
}

#if defined(NDT_SOLVER_STAGE_PROF) && defined(__HIP_DEVICE_COMPILE__)
// stage clocks of the solver (profiling builds of ndt_match.hip, which defines g_solver_prof): [2k] cycles, [2k + 1] calls of stage k
#define NDT_STAGE(k, call) { long long t0_ = __builtin_readcyclecounter(); call; atomicAdd((unsigned long long *)&g_solver_prof[2 * (k)], (unsigned long long)((long long)__builtin_readcyclecounter() - t0_)); atomicAdd((unsigned long long *)&g_solver_prof[2 * (k) + 1], 1ull); }
#else
#define NDT_STAGE(k, call) { call; }
#endif

// one Newton iteration up to the start of the line search: what the caller runs next (NEXT_NONE: the registration is over)
NDT_HD int newton_solve(MatchState &st, const double *sums, const double *fsums, const NdtMatchParamsDev &prm, NewtonWs &ws)
{
    NDT_STAGE(0, newton_assemble(st, sums, fsums, prm, ws))
    if (st.use_tikhonov) newton_tikhonov(st, prm, ws);
    if (ws.gnorm <= prm.delta_score) {     // fusion.h:943-965 (the regularisation before it has no other effect)
        if (st.score_here > st.score_best) st.T = st.Tbest;
        st.exit_code = 1;
        st.done = 1;
        return NEXT_NONE;
    }
    NDT_STAGE(1, newton_factor(ws))
    if (!ws.is_pd) {
        NDT_STAGE(2, newton_regularize(ws))
        NDT_STAGE(3, newton_ldlt(ws))
    }
    int next;
    NDT_STAGE(4, next = newton_finish(st, sums, prm, ws))
    return next;
}

NDT_HD void newton_step(MatchState &st, const double *sums, const NdtMatchParamsDev &prm, NewtonWs &ws)
{
    const int next = newton_solve(st, sums, nullptr, prm, ws);
    if (next == NEXT_APPLY_STEP) NDT_STAGE(5, apply_step(st, prm))
    else if (next == NEXT_REQUEST_TRIAL) NDT_STAGE(6, mt_request_trial(st))
}

// tail of the More-Thuente while(1) body after the trial evaluation (fusion.h:637-790)
// last evaluation at the returned pose (fusion.h:1085-1121)
NDT_HD void match_state_final(MatchState &st, const double *sums, const double *fsums = nullptr)
{
    st.fevals++;
    st.score_here = sums[0];
    if (st.use_feat) st.score_here += fsums[0];           // fusion.h:1085-1096
    if (st.use_prior) st.score_here += prior_score(st);   // fusion.h:1098-1110
    if (st.use_tikhonov) st.score_here += tikhonov_score(st);   // fusion.h:1113-1115: x0 of the LAST Newton evaluation
    if (st.score_here > st.score_best) st.T = st.Tbest;
    st.done = 1;
}

NDT_HDN int linesearch_step(MatchState &st, const double *sums, const NdtMatchParamsDev &prm)
{
    const double ftol = 0.11111, gtol = 0.99999, stpmax = 4.0, stpmin = 0.001, xtol = 0.01, recoverystep = 0.1;
    const int maxfev = 40;
    MTState m = st.mt;
    st.fevals++;
    double f = sums[0] + st.ls_fconst;
    double dg = 0;
    for (int a = 0; a < 6; a++) dg += st.incr[a] * (sums[1 + a] + st.ls_gconst[a]);
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

        // sums hold the Hessian at the accepted pose: match_state_step consumes them once more (unless apply_step ends
        // the registration)
        st.reuse_sums = (info == 1 && st.trial_has_h) ? 1 : 0;
        st.spec_ok = first_accepted ? 1 : 0;
        st.final_from_trial = (info == 1) ? 1 : 0;
        st.mt = m;
        st.step_size = (info == 1) ? m.stp : recoverystep;
        return NEXT_APPLY_STEP;
    }
    if (m.stage1 && (f <= ftest1) && (dg >= dmin(ftol, gtol) * m.dginit)) m.stage1 = 0;
    // (fusion.h:737-776: in the first stage the modified function f - stp * dgtest is handed to cstep.  Both cases go
    //  through the same by-value call: selecting between the addresses of m.fx / fxm would put the block on the stack.)
    {
        const bool mod = m.stage1 && (f <= m.fx) && (f > ftest1);
        double fp = f, dp = dg, fxm = m.fx, fym = m.fy, dgxm = m.dgx, dgym = m.dgy;
        if (mod) {
            fp = f - m.stp * m.dgtest;
            fxm = m.fx - m.stx * m.dgtest;
            fym = m.fy - m.sty * m.dgtest;
            dp = dg - m.dgtest;
            dgxm = m.dgx - m.dgtest;
            dgym = m.dgy - m.dgtest;
        }
        m.infoc = mt_cstep(m.stx, fxm, dgxm, m.sty, fym, dgym, m.stp, fp, dp, m.brackt, m.stmin, m.stmax);
        if (mod) {
            fxm = fxm + m.stx * m.dgtest;
            fym = fym + m.sty * m.dgtest;
            dgxm = dgxm + m.dgtest;
            dgym = dgym + m.dgtest;
        }
        m.fx = fxm; m.fy = fym; m.dgx = dgxm; m.dgy = dgym;
    }
    if (m.brackt) {
        if (fabs(m.sty - m.stx) >= 0.66 * m.width1) m.stp = m.stx + 0.5 * (m.sty - m.stx);
        m.width1 = m.width;
        m.width = fabs(m.sty - m.stx);
    }
    st.mt = m;
    return NEXT_REQUEST_TRIAL;
}


// T0: column-major 4x4 (Eigen::Affine3d storage) or NULL for identity
// Q36: Tcov^-1 row-major, or NULL for NDTMatcherD2D::match; prm.fusion_flags says what it is used for
NDT_HD void match_state_init(MatchState &st, const double *T16, const NdtMatchParamsDev &prm, const double *Q36 = nullptr)
{
    st.use_prior = Q36 != nullptr && (prm.fusion_flags & 1);
    st.use_tikhonov = Q36 != nullptr && (prm.fusion_flags & 2);
    st.use_feat = 0; st.step_ndt = 0.0; st.fevals_saved = 0; st.pad_feat = 0;     // (set by the caller that has feature maps)
    st.ls_joint = 0; st.ls_fconst = 0.0;
    for (int a = 0; a < 6; a++) st.ls_gconst[a] = 0.0;
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
    st.spec_itr_max = prm.itr_max; st.final_from_trial = 0; st.spec_delta = prm.delta_score;
    if ((prm.dof_mask & 0x3f) == 0 || prm.n_neighbours < 0 || prm.n_neighbours > 3) { st.done = 1; st.ret = 0; st.exit_code = -1; }
}

// consumes the sums of the evaluation that was requested (sums[0] score, [1..6] gradient, [7..27] upper
// triangle of the Hessian) and either requests the next evaluation or finishes
NDT_HD void match_state_step(MatchState &st, const double *sums, const NdtMatchParamsDev &prm, NewtonWs &ws)
{
    if (st.phase == PH_LS_TRIAL) {
        st.reuse_sums = 0;
        int next;
        NDT_STAGE(7, next = linesearch_step(st, sums, prm))
        if (next == NEXT_REQUEST_TRIAL) { NDT_STAGE(6, mt_request_trial(st)) return; }
        NDT_STAGE(5, apply_step(st, prm))
        if (st.phase == PH_FINAL && st.final_from_trial && !st.use_feat) { match_state_final(st, sums); return; }
        // an accepted first trial that was evaluated with its Hessian: the evaluation apply_step just requested (the
        // next Newton iteration's, or the final one) is the one these sums come from (same cells, same pose)
        if (!st.reuse_sums) return;
    }
    if (st.phase == PH_NEWTON) newton_step(st, sums, prm, ws);
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
