// ndt_match.hip -- D2D-NDT matcher on CDNA4 (gfx950): K4 derivatives + K5 Newton / More-Thuente,
// one PERSISTENT workgroup per scan pair (K6 batch driver = the grid).
//
// Replaces (reference call sites; perception_oru semantics per SURVEY.md App. A.4-A.6):
//   NDTMatcherD2D::match(target, source, T, true)          ndt_feature/src/ndt_feature_src/ndt_feature_graph.cpp:273
//   NDTMatcherD2D_2D::match                                 ndt_feature/include/ndt_feature/ndt_matcher_d2d_fusion.h:1175
//   NDTMatcherD2D::derivativesNDT / lineSearchMT / MoreThuente::cstep
//                                                           ...ndt_matcher_d2d_fusion.h:856, 1013, 756, 775
// whose Newton loop and line-search driver are restated from the in-repo copy
//   ndt_matcher_d2d_fusion.h:847-1121 (loop), :390-793 (More-Thuente driver, constants :400-408).
//
// Design (DESIGN.md "Match kernel"):
//   * one 256-thread workgroup runs the WHOLE registration of one pair: every derivative evaluation,
//     the 6x6 eigen-regularisation, LDL^T solve, the More-Thuente state machine and the final
//     best-score rollback, so ~25-130 dependent evaluations cost zero host round trips and pairs that
//     converge early simply free their CU (no lock-step over the batch).
//   * the source cells are never copied or re-written: each evaluation applies the composed pose
//     (trial step x current pose) to the original 80-byte records (the reference re-allocates every
//     cell per trial, fusion.h:563-589).
//   * an evaluation is two interleaved wave-level stages: PROBE (each lane owns a source cell, looks up
//     the (2n+1)^3 dense-table slots around its transformed mean, 25 independent loads in flight per
//     z-layer) pushes hits into a per-wave LDS queue by ballot/popcount compaction; TERM pops 64
//     (source, target) pairs so that all 64 lanes do the ~0.2-0.6 kFLOP fp64 pair term densely.
//     No divergence in the heavy math, no atomics, fixed summation order -> run-to-run identical.
//   * 1+6+21 fp64 partial sums per lane -> wave xor-shuffle tree -> 4 LDS partials -> fixed-order sum.
//   * MFMA is not used: nothing here is a dense contraction (3x3 / 6x6 per-pair expressions).
#include "ndt_math.h"
#include <float.h>

#define NDT_MATCH_THREADS 512
#define NDT_MATCH_WAVES (NDT_MATCH_THREADS / 64)
#define NDT_QN 512

namespace {

struct MapView {
    const int32_t *table;
    const NdtCell *cells;
    int n_cells;
    int sx, sy, sz;
    double cx, cy, cz, res;
};

NDT_D MapView map_view(const NdtSetView &s, unsigned map)
{
    MapView v;
    v.table = s.table + (size_t)map * s.grid.slots;
    v.cells = s.cells + (size_t)map * s.grid.max_cells;
    v.n_cells = (int)s.counters[map].n_cells;
    v.sx = s.grid.size[0]; v.sy = s.grid.size[1]; v.sz = s.grid.size[2];
    v.cx = s.centres[map * 3]; v.cy = s.centres[map * 3 + 1]; v.cz = s.centres[map * 3 + 2];
    v.res = s.grid.res;
    return v;
}

NDT_D double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

NDT_D unsigned long long lanemask_lt()
{
    unsigned lane = threadIdx.x & 63u;
    return (lane == 0) ? 0ull : (~0ull >> (64u - lane));
}

// One (source cell, target cell) term of NDTMatcherD2D::derivativesNDT + updateGradientHessianLocal
// (SURVEY.md App. A.4).  m, C: source mean / covariance already in the target frame.
// acc: [0] score, [1..6] gradient, [7..27] upper triangle of the Hessian (row-major).
template <bool WITH_H>
NDT_D void pair_term(d3 m, sym3 C, d3 mu, sym3 Cj, double lfd1, double lfd2, double *acc)
{
    d3 x = m - mu;
    sym3 B;
    if (!inverse_check(C + Cj, B)) return;          // CSum.computeInverseAndDetWithCheck
    d3 xB = mul(B, x);
    double l = dot(x, xB);
    if (!(l * 0.0 == 0.0)) return;                   // if(l*0 != 0) continue;
    double sh = -lfd1 * exp(-lfd2 * l * 0.5);
    double f = -(lfd2 * 0.5) * sh;
    d3 w = mul(C, xB);
    d3 c = cross(w, xB);                             // x^T B Z_k B x = 2 c_k
    d3 d = cross(m, xB);                             // x^T B j_k     = d_k   (j_k = e_k x m)
    double Q[6] = {2.0 * xB.x, 2.0 * xB.y, 2.0 * xB.z, 2.0 * (d.x - c.x), 2.0 * (d.y - c.y), 2.0 * (d.z - c.z)};
    acc[0] += sh;
#pragma unroll
    for (int a = 0; a < 6; a++) acc[1 + a] += f * Q[a];
    if (!WITH_H) return;

    const double kq = lfd2 * 0.5;
    d3 j[3] = {ex_cross(m), ey_cross(m), ez_cross(m)};
    d3 Bj[3], p[3], r[3], u[3];
    p[0] = ex_cross(xB); p[1] = ey_cross(xB); p[2] = ez_cross(xB);
    d3 qv[3] = {ex_cross(w), ey_cross(w), ez_cross(w)};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        Bj[k] = mul(B, j[k]);
        r[k] = qv[k] - mul(C, p[k]);                 // Z_k (B x)
        u[k] = mul(B, r[k]);
    }
    const double Bm[3][3] = {{B.xx, B.xy, B.xz}, {B.xy, B.yy, B.yz}, {B.xz, B.yz, B.zz}};
    const double Bjv[3][3] = {{Bj[0].x, Bj[0].y, Bj[0].z}, {Bj[1].x, Bj[1].y, Bj[1].z}, {Bj[2].x, Bj[2].y, Bj[2].z}};
    const double uv[3][3] = {{u[0].x, u[0].y, u[0].z}, {u[1].x, u[1].y, u[1].z}, {u[2].x, u[2].y, u[2].z}};
    // 2 x^T B H_ik, H_ik = e_i x (e_k x m), i <= k
    const double xBH[3][3] = {{-2.0 * (xB.y * m.y + xB.z * m.z), 2.0 * xB.y * m.x, 2.0 * xB.z * m.x},
                              {0.0, -2.0 * (xB.x * m.x + xB.z * m.z), 2.0 * xB.z * m.y},
                              {0.0, 0.0, -2.0 * (xB.x * m.x + xB.y * m.y)}};
    int o = 7;
#pragma unroll
    for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int b = a; b < 6; b++) {
            double h;
            if (b < 3) {
                h = 2.0 * Bm[a][b];
            } else if (a < 3) {
                h = 2.0 * Bjv[b - 3][a] - 2.0 * uv[b - 3][a];
            } else {
                int i = a - 3, k = b - 3;
                h = 2.0 * dot(j[i], Bj[k]) + xBH[i][k] - 2.0 * (dot(u[i], j[k]) + dot(u[k], j[i])) +
                    2.0 * dot(u[i], r[k]) + 2.0 * dot(p[i], r[k]);
            }
            acc[o++] += f * (h - kq * Q[a] * Q[b]);
        }
    }
}

// One evaluation of derivativesNDT over all source cells, transformed by T.  All threads of the
// workgroup participate; result in s_sums[0..6] (and [7..27] when WITH_H).  Ends with a barrier.
template <int NN, bool WITH_H>
NDT_D void eval_derivs(const MapView &tg, const NdtCell *__restrict__ src, int msrc, const rigid &T, double lfd1,
                       double lfd2, double *s_src, uint32_t *s_queue, double *s_part, double *s_sums)
{
    constexpr int NACC = WITH_H ? 28 : 7;
    constexpr int W = 2 * NN + 1;
    static_assert(63 + W * 64 <= NDT_QN, "per-wave hit queue too small for this neighbourhood");
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; k++) acc[k] = 0.0;
    const unsigned tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    double *mysrc = s_src + wave * (9 * 64);
    uint32_t *myq = s_queue + wave * NDT_QN;
    unsigned qhead = 0, qcount = 0;   // wave-uniform
    const unsigned long long lt = lanemask_lt();

    // TERM stage: pops up to 64 (source lane, target cell) pairs; every lane does one dense pair term.
    // `min_fill` = 64 while probing (only full batches), 1 for the final flush of a source tile.
    auto drain = [&](unsigned min_fill) {
        while (qcount >= min_fill && qcount > 0) {
            unsigned n = qcount < 64u ? qcount : 64u;
            if (lane < n) {
                uint32_t e = myq[(qhead + lane) & (NDT_QN - 1)];
                unsigned sl = e >> 24, id = e & 0xFFFFFFu;
                d3 m = {mysrc[0 * 64 + sl], mysrc[1 * 64 + sl], mysrc[2 * 64 + sl]};
                sym3 C = {mysrc[3 * 64 + sl], mysrc[4 * 64 + sl], mysrc[5 * 64 + sl],
                          mysrc[6 * 64 + sl], mysrc[7 * 64 + sl], mysrc[8 * 64 + sl]};
                const NdtCell *tc = tg.cells + id;
                d3 mu = {tc->mean[0], tc->mean[1], tc->mean[2]};
                sym3 Cj = {tc->cov[0], tc->cov[1], tc->cov[2], tc->cov[3], tc->cov[4], tc->cov[5]};
                pair_term<WITH_H>(m, C, mu, Cj, lfd1, lfd2, acc);
            }
            qhead = (qhead + n) & (NDT_QN - 1);
            qcount -= n;
        }
    };

    // every wave owns a contiguous, equally sized range of source cells (even a small map keeps all
    // waves busy: 372 cells -> 47 per wave instead of 6 full waves + 2 idle ones)
    const int per_wave = (msrc + NDT_MATCH_WAVES - 1) / NDT_MATCH_WAVES;
    const int w_begin = (int)wave * per_wave, w_end = min(msrc, w_begin + per_wave);
    for (int base = w_begin; base < w_end; base += 64) {
        int i = base + (int)lane;
        bool vi = i < w_end;
        int ix = 0, iy = 0, iz = 0;
        if (vi) {
            const NdtCell *sc = src + i;
            d3 m0 = {sc->mean[0], sc->mean[1], sc->mean[2]};
            sym3 C0 = {sc->cov[0], sc->cov[1], sc->cov[2], sc->cov[3], sc->cov[4], sc->cov[5]};
            d3 m = apply(T, m0);                    // pseudoTransformNDT: mean' = T mean
            sym3 C = rotate_cov(T.r, C0);           //                      cov'  = R cov R^T
            mysrc[0 * 64 + lane] = m.x; mysrc[1 * 64 + lane] = m.y; mysrc[2 * 64 + lane] = m.z;
            mysrc[3 * 64 + lane] = C.xx; mysrc[4 * 64 + lane] = C.xy; mysrc[5 * 64 + lane] = C.xz;
            mysrc[6 * 64 + lane] = C.yy; mysrc[7 * 64 + lane] = C.yz; mysrc[8 * 64 + lane] = C.zz;
            ix = lazygrid_index(m.x, tg.cx, tg.res, tg.sx);   // getCellsForPoint(mean, n_neighbours)
            iy = lazygrid_index(m.y, tg.cy, tg.res, tg.sy);
            iz = lazygrid_index(m.z, tg.cz, tg.res, tg.sz);
        }
        // PROBE stage: (2n+1)^3 dense-table slots around the lane's cell; one row of W loads in flight
        for (int dz = -NN; dz <= NN; dz++) {
            int zz = iz + dz;
            bool zok = vi && zz >= 0 && zz < tg.sz;
            if (!__ballot(zok)) continue;
#pragma unroll 1
            for (int dx = -NN; dx <= NN; dx++) {
                int xx = ix + dx;
                bool xok = zok && xx >= 0 && xx < tg.sx;
                int ids[W];
#pragma unroll
                for (int dy = 0; dy < W; dy++) {
                    int yy = iy + dy - NN;
                    bool ok = xok && yy >= 0 && yy < tg.sy;
                    ids[dy] = ok ? tg.table[((size_t)xx * tg.sy + yy) * tg.sz + zz] : -1;
                }
#pragma unroll
                for (int k = 0; k < W; k++) {
                    bool hit = ids[k] >= 0;
                    unsigned long long mask = __ballot(hit);
                    if (hit)
                        myq[(qhead + qcount + (unsigned)__popcll(mask & lt)) & (NDT_QN - 1)] =
                            (lane << 24) | (uint32_t)ids[k];
                    qcount += (unsigned)__popcll(mask);
                }
                drain(64);
            }
        }
        drain(1);   // the per-wave source tile is overwritten by the next batch
    }

    // 28 (or 7) sums: wave tree, then fixed-order sum of the wave partials
#pragma unroll
    for (int k = 0; k < NACC; k++) {
        double v = wave_sum_d(acc[k]);
        if (lane == 0) s_part[wave * 28 + k] = v;
    }
    __syncthreads();
    if (tid < (unsigned)NACC) {
        double s = 0;
        for (int w = 0; w < NDT_MATCH_WAVES; w++) s += s_part[w * 28 + tid];
        s_sums[tid] = s;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// serial part (thread 0): Newton step + More-Thuente state machine
// ------------------------------------------------------------------------------------------------
enum { PH_NEWTON = 0, PH_LS_TRIAL = 1, PH_FINAL = 2 };

struct MatchState {
    rigid T, Tbest, Teval;
    double score_best, score_here;
    double incr[6];
    // More-Thuente (fusion.h:400-408, 485-521)
    double stp, finit, dginit, dgtest, width, width1, stx, fx, dgx, sty, fy, dgy, stmin, stmax;
    int brackt, stage1, nfev, infoc;
    int itr_ctr, fevals, ret, exit_code, phase, with_h, done;
    // workspace of the pivoted LDL^T (dynamically indexed -> kept in LDS, not in scratch)
    double ws_a[36], ws_y[6], ws_H[36], ws_g[6], ws_dx[6];
    int ws_perm[6];
};

NDT_D double dmin(double a, double b) { return a < b ? a : b; }
NDT_D double dmax(double a, double b) { return a > b ? a : b; }
NDT_D double absmax3(double a, double b, double c) { return dmax(dmax(fabs(a), fabs(b)), fabs(c)); }

// MoreThuente::cstep = MINPACK mcstep (published algorithm: More & Thuente, ACM TOMS 20(3), 1994);
// call sites fusion.h:756,775.
__device__ __noinline__ int mt_cstep(double &stx, double &fx, double &dx, double &sty, double &fy, double &dy,
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
__device__ __noinline__ void mt_request_trial(MatchState &st)
{
    const double stpmax = 4.0, stpmin = 0.001, xtol = 0.01;
    const int maxfev = 40;
    if (st.brackt) {
        st.stmin = dmin(st.stx, st.sty);
        st.stmax = dmax(st.stx, st.sty);
    } else {
        st.stmin = st.stx;
        st.stmax = st.stp + 4 * (st.stp - st.stx);
    }
    st.stp = dmax(st.stp, stpmin);
    st.stp = dmin(st.stp, stpmax);
    if ((st.brackt && ((st.stp <= st.stmin) || (st.stp >= st.stmax))) || (st.nfev >= maxfev - 1) ||
        (st.infoc == 0) || (st.brackt && (st.stmax - st.stmin <= xtol * st.stmax)))
        st.stp = st.stx;
    double pincr[6];
    for (int a = 0; a < 6; a++) pincr[a] = st.stp * st.incr[a];
    rigid ps;
    pose_to_rigid(pincr, ps);
    rigid_mul(ps, st.T, st.Teval);     // trial cells = ps * nextNDT (fusion.h:556-589)
    st.with_h = 0;
    st.phase = PH_LS_TRIAL;
}

// pose update + convergence tests (fusion.h:1032-1080)
__device__ __noinline__ void apply_step(MatchState &st, double step_size, const NdtMatchParamsDev &prm)
{
    double inorm = 0;
    for (int a = 0; a < 6; a++) {
        st.incr[a] *= step_size;
        inorm += st.incr[a] * st.incr[a];
    }
    inorm = sqrt(inorm);
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

__device__ __noinline__ void newton_step(MatchState &st, const double *sums, const NdtMatchParamsDev &prm)
{
    st.fevals++;
    st.score_here = sums[0];
    if (st.score_here < st.score_best) {   // fusion.h:914-920
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
    double pad = 0.0;
    bool havepad = false;
    double gnorm = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) {
        bool on = (prm.dof_mask >> a) & 1;
        g[a] = on ? sums[1 + a] : 0.0;
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
    // H = V diag(evals) V^T, i.e. H + regularizer*I: only lambda_min and lambda_max are needed.  A
    // positive definite H (the usual case near the optimum) is certified by an unpivoted Cholesky
    // and skips the eigen-decomposition altogether.
    if (!chol_is_pd<6>(H)) {
        double A[6][6], V[6][6];
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = 0; b < 6; b++) A[a][b] = H[a][b];
        jacobi_static<6, false>(A, V);
        double minC = A[0][0], maxC = A[0][0];
#pragma unroll
        for (int a = 1; a < 6; a++) { minC = dmin(minC, A[a][a]); maxC = dmax(maxC, A[a][a]); }
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
#pragma unroll
    for (int a = 0; a < 6; a++) {
        st.ws_g[a] = g[a];
#pragma unroll
        for (int b = 0; b < 6; b++) st.ws_H[a * 6 + b] = H[a][b];
    }
    ldlt_solve_ws<6>(6, st.ws_H, st.ws_g, st.ws_dx, st.ws_a, st.ws_y, st.ws_perm);
    double dginit = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) {
        bool on = (prm.dof_mask >> a) & 1;
        double d = on ? -st.ws_dx[a] : 0.0;
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
    st.finit = st.score_here;
    st.dginit = dginit;
    if (st.dginit >= 0.0) {                // fusion.h:456-479
        for (int a = 0; a < 6; a++) st.incr[a] = -st.incr[a];
        st.dginit = -st.dginit;
        if (st.dginit >= 0.0) {
            apply_step(st, 0.1, prm);
            return;
        }
    }
    st.stp = 1.0;
    st.brackt = 0; st.stage1 = 1; st.nfev = 0; st.infoc = 1;
    st.dgtest = 0.11111 * st.dginit;
    st.width = 4.0 - 0.001;
    st.width1 = 2 * st.width;
    st.stx = 0.0; st.fx = st.finit; st.dgx = st.dginit;
    st.sty = 0.0; st.fy = st.finit; st.dgy = st.dginit;
    mt_request_trial(st);
}

// tail of the More-Thuente while(1) body after the trial evaluation (fusion.h:637-790)
__device__ __noinline__ void linesearch_step(MatchState &st, const double *sums, const NdtMatchParamsDev &prm)
{
    const double ftol = 0.11111, gtol = 0.99999, stpmax = 4.0, stpmin = 0.001, xtol = 0.01, recoverystep = 0.1;
    const int maxfev = 40;
    st.fevals++;
    double f = sums[0];
    double dg = 0;
    for (int a = 0; a < 6; a++) dg += st.incr[a] * sums[1 + a];
    st.nfev++;
    double ftest1 = st.finit + st.stp * st.dgtest;
    int info = 0;
    if ((st.brackt && ((st.stp <= st.stmin) || (st.stp >= st.stmax))) || (st.infoc == 0)) info = 6;
    if ((st.stp == stpmax) && (f <= ftest1) && (dg <= st.dgtest)) info = 5;
    if ((st.stp == stpmin) && ((f > ftest1) || (dg >= st.dgtest))) info = 4;
    if (st.nfev >= maxfev) info = 3;
    if (st.brackt && (st.stmax - st.stmin <= xtol * st.stmax)) info = 2;
    if ((f <= ftest1) && (fabs(dg) <= gtol * (-st.dginit))) info = 1;
    if (info != 0) {
        apply_step(st, (info == 1) ? st.stp : recoverystep, prm);
        return;
    }
    if (st.stage1 && (f <= ftest1) && (dg >= dmin(ftol, gtol) * st.dginit)) st.stage1 = 0;
    if (st.stage1 && (f <= st.fx) && (f > ftest1)) {
        double fm = f - st.stp * st.dgtest;
        double fxm = st.fx - st.stx * st.dgtest;
        double fym = st.fy - st.sty * st.dgtest;
        double dgm = dg - st.dgtest;
        double dgxm = st.dgx - st.dgtest;
        double dgym = st.dgy - st.dgtest;
        st.infoc = mt_cstep(st.stx, fxm, dgxm, st.sty, fym, dgym, st.stp, fm, dgm, st.brackt, st.stmin, st.stmax);
        st.fx = fxm + st.stx * st.dgtest;
        st.fy = fym + st.sty * st.dgtest;
        st.dgx = dgxm + st.dgtest;
        st.dgy = dgym + st.dgtest;
    } else {
        st.infoc = mt_cstep(st.stx, st.fx, st.dgx, st.sty, st.fy, st.dgy, st.stp, f, dg, st.brackt, st.stmin, st.stmax);
    }
    if (st.brackt) {
        if (fabs(st.sty - st.stx) >= 0.66 * st.width1) st.stp = st.stx + 0.5 * (st.sty - st.stx);
        st.width1 = st.width;
        st.width = fabs(st.sty - st.stx);
    }
    mt_request_trial(st);
}

}  // namespace

template <int NN>
__global__ __launch_bounds__(NDT_MATCH_THREADS) void ndt_match_kernel(
    NdtSetView tset, const uint32_t *__restrict__ tidx, NdtSetView sset, const uint32_t *__restrict__ sidx,
    double *__restrict__ T16, NdtMatchParamsDev prm, NdtMatchResultDev *__restrict__ res)
{
    __shared__ double s_src[NDT_MATCH_WAVES * 9 * 64];
    __shared__ uint32_t s_queue[NDT_MATCH_WAVES * NDT_QN];
    __shared__ double s_part[NDT_MATCH_WAVES * 28];
    __shared__ double s_sums[28];
    __shared__ MatchState st;

    const unsigned pair = blockIdx.x;
    const MapView tg = map_view(tset, tidx[pair]);
    const MapView sv = map_view(sset, sidx[pair]);
    double *Tio = T16 + (size_t)pair * 16;

    if (threadIdx.x == 0) {
        rigid T0;
        if (prm.use_initial_guess) {
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) T0.r[r * 3 + c] = Tio[c * 4 + r];   // column-major Affine3d
                T0.t[r] = Tio[12 + r];
            }
        } else {
            for (int k = 0; k < 9; k++) T0.r[k] = (k % 4 == 0) ? 1.0 : 0.0;
            T0.t[0] = T0.t[1] = T0.t[2] = 0.0;
        }
        st.T = T0; st.Tbest = T0; st.Teval = T0;
        st.score_best = DBL_MAX; st.score_here = 0;
        st.itr_ctr = 0; st.fevals = 0; st.ret = 1; st.exit_code = 0;
        st.phase = PH_NEWTON; st.with_h = 1; st.done = 0;
        if ((prm.dof_mask & 0x3f) == 0 || prm.n_neighbours < 0 || prm.n_neighbours > 3) { st.done = 1; st.ret = 0; st.exit_code = -1; }
    }
    __syncthreads();

    long long cyc_eval = 0, cyc_solver = 0;
    while (!st.done) {
        const rigid Te = st.Teval;
        const int with_h = st.with_h;
        __syncthreads();   // everyone has read the request before thread 0 may rewrite it
        long long c0 = __builtin_readcyclecounter();
        if (with_h) eval_derivs<NN, true>(tg, sv.cells, sv.n_cells, Te, prm.lfd1, prm.lfd2, s_src, s_queue, s_part, s_sums);
        else eval_derivs<NN, false>(tg, sv.cells, sv.n_cells, Te, prm.lfd1, prm.lfd2, s_src, s_queue, s_part, s_sums);
        long long c1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) {
            if (st.phase == PH_NEWTON) newton_step(st, s_sums, prm);
            else if (st.phase == PH_LS_TRIAL) linesearch_step(st, s_sums, prm);
            else {   // PH_FINAL: fusion.h:1085-1121
                st.fevals++;
                st.score_here = s_sums[0];
                if (st.score_here > st.score_best) st.T = st.Tbest;
                st.done = 1;
            }
            cyc_eval += c1 - c0;
            cyc_solver += (long long)__builtin_readcyclecounter() - c1;
        }
        __syncthreads();
    }

    if (threadIdx.x == 0) {
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) Tio[c * 4 + r] = st.T.r[r * 3 + c];
            Tio[12 + r] = st.T.t[r];
            Tio[r * 4 + 3] = 0.0;
        }
        Tio[15] = 1.0;
        NdtMatchResultDev o;
        o.converged = st.ret;
        o.iterations = st.itr_ctr;
        o.fevals = st.fevals;
        o.exit_code = st.exit_code;
        o.score = (st.score_here > st.score_best) ? st.score_best : st.score_here;
        o.n_source = sv.n_cells;
        o.n_target = tg.n_cells;
        o.cycles_eval = cyc_eval;
        o.cycles_solver = cyc_solver;
        res[pair] = o;
    }
}

// NDTMatcherD2D::derivativesNDT as a stand-alone entry (host-driven matchFusion loop, FD tests).
template <int NN>
__global__ __launch_bounds__(NDT_MATCH_THREADS) void ndt_derivatives_kernel(
    NdtSetView tset, unsigned tmap, const NdtCell *__restrict__ src, unsigned m, int with_h, double lfd1,
    double lfd2, double *__restrict__ out28)
{
    __shared__ double s_src[NDT_MATCH_WAVES * 9 * 64];
    __shared__ uint32_t s_queue[NDT_MATCH_WAVES * NDT_QN];
    __shared__ double s_part[NDT_MATCH_WAVES * 28];
    __shared__ double s_sums[28];
    const MapView tg = map_view(tset, tmap);
    rigid I;
    for (int k = 0; k < 9; k++) I.r[k] = (k % 4 == 0) ? 1.0 : 0.0;
    I.t[0] = I.t[1] = I.t[2] = 0.0;
    if (threadIdx.x < 28) s_sums[threadIdx.x] = 0.0;
    __syncthreads();
    if (with_h) eval_derivs<NN, true>(tg, src, (int)m, I, lfd1, lfd2, s_src, s_queue, s_part, s_sums);
    else eval_derivs<NN, false>(tg, src, (int)m, I, lfd1, lfd2, s_src, s_queue, s_part, s_sums);
    if (threadIdx.x < 28) out28[threadIdx.x] = s_sums[threadIdx.x];
}

hipError_t ndt_launch_match(const NdtSetView &tset, const uint32_t *tidx_dev, const NdtSetView &sset,
                            const uint32_t *sidx_dev, double *T16_dev, size_t n_pairs, const NdtMatchParamsDev &prm,
                            NdtMatchResultDev *res_dev, hipStream_t stream)
{
    if (n_pairs == 0) return hipSuccess;
#define NDT_LAUNCH_MATCH(NN)                                                                                         \
    hipLaunchKernelGGL(ndt_match_kernel<NN>, dim3((unsigned)n_pairs), dim3(NDT_MATCH_THREADS), 0, stream, tset,      \
                       tidx_dev, sset, sidx_dev, T16_dev, prm, res_dev)
    switch (prm.n_neighbours) {
    case 0: NDT_LAUNCH_MATCH(0); break;
    case 1: NDT_LAUNCH_MATCH(1); break;
    case 2: NDT_LAUNCH_MATCH(2); break;
    case 3: NDT_LAUNCH_MATCH(3); break;
    default: return hipErrorInvalidValue;
    }
#undef NDT_LAUNCH_MATCH
    return hipGetLastError();
}

hipError_t ndt_launch_derivatives(const NdtSetView &tset, size_t tmap, const NdtCell *src_cells_dev, size_t m,
                                  int n_neighbours, int compute_hessian, double lfd1, double lfd2, double *out28_dev,
                                  hipStream_t stream)
{
#define NDT_LAUNCH_DERIV(NN)                                                                                         \
    hipLaunchKernelGGL(ndt_derivatives_kernel<NN>, dim3(1), dim3(NDT_MATCH_THREADS), 0, stream, tset,                \
                       (unsigned)tmap, src_cells_dev, (unsigned)m, compute_hessian, lfd1, lfd2, out28_dev)
    switch (n_neighbours) {
    case 0: NDT_LAUNCH_DERIV(0); break;
    case 1: NDT_LAUNCH_DERIV(1); break;
    case 2: NDT_LAUNCH_DERIV(2); break;
    case 3: NDT_LAUNCH_DERIV(3); break;
    default: return hipErrorInvalidValue;
    }
#undef NDT_LAUNCH_DERIV
    return hipGetLastError();
}
