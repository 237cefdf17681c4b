// ndt_match.hip -- D2D-NDT matcher on CDNA4 (gfx950): K4 derivatives + K5 Newton / More-Thuente,
// PERSISTENT workgroups, one scan pair at a time each (K6 batch driver = the grid).
//
// Replaces (reference call sites; perception_oru semantics per SURVEY.md App. A.4-A.6):
//   NDTMatcherD2D::match(target, source, T, true)          ndt_feature/src/ndt_feature_src/ndt_feature_graph.cpp:273
//   NDTMatcherD2D_2D::match                                 ndt_feature/include/ndt_feature/ndt_matcher_d2d_fusion.h:1175
//   NDTMatcherD2D::derivativesNDT / lineSearchMT / MoreThuente::cstep
//                                                           ...ndt_matcher_d2d_fusion.h:856, 1013, 756, 775
// whose Newton loop and line-search driver are restated from the in-repo copy
//   ndt_matcher_d2d_fusion.h:847-1121 (loop), :390-793 (More-Thuente driver, constants :400-408).
//
// Design:
//   * the WHOLE registration runs on the device: every derivative evaluation, the 6x6 regularisation, the solve,
//     the More-Thuente state machine and the best-score rollback (csrc/ndt_solver.h) -- zero host round trips.
//   * the source cells are never copied or re-written: each evaluation applies the composed pose
//     (trial step x current pose) to the original 80-byte records (the reference re-allocates every
//     cell per trial, fusion.h:563-589).
//   * a workgroup runs a registration.  The source cells are dealt to 8 SHARES (cell i -> share i mod 8); a wave of a
//     wide (8-wave) workgroup sums one share, a wave of a narrow (4-wave) workgroup two.  Per share and group of 64
//     cells: PROBE (each lane owns a source cell and reads the (2n+1)^3 neighbourhood of its transformed mean as bit
//     windows of the target's rank bitmap), every lane counts its hits, a wave scan places them in the share's hit
//     list (kept in LDS and reused while no mean leaves its target cell); TERM takes the list 64 (source, target)
//     pairs at a time so that all 64 lanes do the fp64 pair term densely (444 instructions with the Hessian, 165
//     without).
//   * 1+6+21 fp64 partial sums per lane -> fixed butterfly over the wave (permlane swaps) -> 8 LDS partials (one per
//     share) -> fixed-order sum: wide and narrow workgroups give the same bits.
//   * persistent workgroups pull pairs from a ticket counter; registrations that run long are parked until every
//     pair has started, then finished side by side (NdtMatchWork below).
//   * no atomics in the sums, fixed summation order -> run-to-run identical.
//   * MFMA is not used: nothing here is a dense contraction (3x3 / 6x6 per-pair expressions).
#include "ndt_math.h"
#include "ndt_wave.h"
#ifdef NDT_MATCH_PROF
__device__ long long g_solver_prof[16];
#define NDT_SOLVER_STAGE_PROF
#endif
#include "ndt_solver.h"
#include <float.h>

#ifndef NDT_MATCH_THREADS
#define NDT_MATCH_THREADS 512
#endif
#define NDT_MATCH_WAVES (NDT_MATCH_THREADS / 64)
#define NDT_VW 8             // shares the source cells of an evaluation are dealt to (= waves of a wide workgroup)

namespace {

// How many of the NDT_VW shares of an evaluation hold cells (the others deliver rows of zeros): a property of the source map alone.
// Maps of 257..448 cells -- the bench's halls hold 372 -- are dealt to 5..7 shares of 52..64 cells instead of eight at 32..56 of
// the 64 lanes: TRANSFORM, PROBE, the hit list and the wave sum are paid per share, and the pair terms of a share come in batches
// of 64 (eight shares of 238 terms: 32 batches, six of 318: 30).  Round 6, measured (bench, 100 steps): 642 k -> 664-671 k
// registrations/s; the isolated launch of 1024 pairs 2.16-2.26 -> 2.08-2.13 ms on the same box (a lone registration's six
// waves share four SIMDs with less contention than eight do).  -DNDT_FIXED_SHARES: eight shares always (A/B).
NDT_D unsigned ndt_shares_of(int msrc)
{
#ifndef NDT_FIXED_SHARES
    if (msrc <= 256) return (unsigned)NDT_VW;          // (small maps keep all eight waves: a lone registration's latency)
    // (Measured and not kept: the number of shares, 5..8, that needs the fewest groups of 64 cells in all -- seven for the cluttered
    //  scene's 1 735 cells, 28 groups instead of 32: its matcher launch alone 9.65 ms against 8.86-9.04 with eight.)
    const unsigned n = ((unsigned)msrc + 63u) / 64u;
    return n < (unsigned)NDT_VW ? n : (unsigned)NDT_VW;
#else
    (void)msrc;
    return (unsigned)NDT_VW;
#endif
}

// cell records and rank bitmaps live in device (global) memory: said in the pointer types, so that pointers that went
// through an LDS copy of a MapView are still dereferenced with global_load, not flat_load
typedef const NdtCell __attribute__((address_space(1))) *gcell_ptr;
typedef const unsigned long long __attribute__((address_space(1))) *grank_ptr;   // uint2 {bits, first rank} as one 8-byte word

struct MapView {
    grank_ptr rankmap;
    gcell_ptr cells;
    int n_cells;
    int sx, sy, sz;
    double cx, cy, cz, res;
    double hx, hy, hz;          // size / 2.0 per axis (the addend of LazyGrid's index formula)
    double inv_res;             // 1 / res when res is a power of two (the quotient of the index formula is then a product, bit for
                                // bit: 3 instructions instead of the 42 of three IEEE divisions per source cell and evaluation), else 0
};

// lazygrid_index with size / 2.0 handed in (a value the large-map kernels keep in scalar registers)
NDT_D int lazygrid_index_h(double p, double centre, double res, double half)
{
#pragma clang fp contract(off)
    double v = floor((p - centre) / res + 0.5) + half;
    if (!(v > -2.0e9 && v < 2.0e9)) return -1;
    return (int)v;
}
// the same for a cell size that is a power of two: x / 2^k == x * 2^-k in every case (both are the correctly rounded value of
// the same real number; no fused multiply-add: the product is rounded before 0.5 is added, like the quotient)
NDT_D int lazygrid_index_p2(double p, double centre, double inv_res, double half)
{
#pragma clang fp contract(off)
    double q = (p - centre) * inv_res;
    double v = floor(q + 0.5) + half;
    if (!(v > -2.0e9 && v < 2.0e9)) return -1;
    return (int)v;
}
NDT_D double pow2_reciprocal(double res)
{
    int e;
    return (res > 0.0 && frexp(res, &e) == 0.5 && e > -1000 && e < 1000) ? 1.0 / res : 0.0;
}

NDT_D MapView map_view(const NdtSetView &s, unsigned map)
{
    MapView v;
    v.rankmap = (grank_ptr)(s.rankmap + (size_t)map * ndt_rm_stride(s.grid));
    v.cells = (gcell_ptr)ndt_cells_of(s, map, s.cell_sel ? s.cell_sel[map] : 0u);   // (second array after an incremental update)
    v.n_cells = (int)s.counters[map].n_cells;
    v.sx = s.grid.size[0]; v.sy = s.grid.size[1]; v.sz = s.grid.size[2];
    v.cx = s.centres[map * 3]; v.cy = s.centres[map * 3 + 1]; v.cz = s.centres[map * 3 + 2];
    v.res = s.grid.res;
    v.inv_res = pow2_reciprocal(s.grid.res);
    v.hx = s.grid.half[0]; v.hy = s.grid.half[1]; v.hz = s.grid.half[2];
    return v;
}

// The same for a wave-uniform map index (the large-map kernels: one registration per workgroup / task): the grid
// geometry moves to SCALAR registers.  The centres arrive through vector loads, and as vector registers they were what
// the <2> variants of those kernels spilled (14 / 26 registers, profiles/r03_kernel_resources.txt).
NDT_D double uniform_d(double v)
{
    // (v_readfirstlane in an asm statement, not the builtin: the optimiser folds the builtin away when it can prove its
    //  operand uniform -- and then keeps the value in the vector register it was computed in)
    int lo, hi;
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(lo) : "v"(__double2loint(v)));
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(hi) : "v"(__double2hiint(v)));
    return __hiloint2double(hi, lo);
}
NDT_D MapView map_view_uniform(const NdtSetView &s, unsigned map)
{
    MapView v = map_view(s, (unsigned)__builtin_amdgcn_readfirstlane((int)map));
    v.n_cells = __builtin_amdgcn_readfirstlane(v.n_cells);
    v.cx = uniform_d(v.cx); v.cy = uniform_d(v.cy); v.cz = uniform_d(v.cz);
    return v;
}

NDT_D double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// (pl_swap_add, xor_lane, wave_sum_step: csrc/ndt_wave.h)
template <int N>
NDT_D double wave_sum_all(double (&v)[N])
{
    static_assert(N == 8 || N == 32, "padded value count");
    const unsigned lane = threadIdx.x & 63u;
    wave_sum_step<N, N / 2, 32>(v, lane);
    double t = v[0];
    if constexpr (N == 8) {          // 8 values: the lane bits 2, 1, 0 are left
        t += xor_lane<4>(t);
        t += xor_lane<2>(t);
        t += xor_lane<1>(t);
    } else {                         // 32 values: lane bit 0 is left
        t += xor_lane<1>(t);
    }
    return t;
}

NDT_D unsigned long long lanemask_lt()
{
    unsigned lane = threadIdx.x & 63u;
    return (lane == 0) ? 0ull : (~0ull >> (64u - lane));
}

// A constant that is (re)made in scalar registers where it is used: the pair term is inlined into loops whose vector
// registers are all taken, and a loop-invariant constant hoisted into a vector register there is spilled to scratch
// memory and reloaded, with a full memory wait, at every use.
NDT_D double sgpr_const(double c)
{
    asm volatile("" : "+s"(c));
    return c;
}

// exp(x) for x <= 0 (the exponent of a Gaussian): 2^k exp(r), k = rint(x log2 e), r = x - k ln 2 in two parts
// (|r| <= 0.3466), exp(r) by its Taylor polynomial of degree 13 (truncation 4e-18), about one ulp like the library's;
// 0 below the double range, NaN for NaN.
NDT_D double exp_nonpos(double x)
{
    const double kf = rint(x * sgpr_const(1.4426950408889634));
    double r = fma(kf, sgpr_const(-6.93147180369123816490e-01), x);
    r = fma(kf, sgpr_const(-1.90821492927058770002e-10), r);
    double p = sgpr_const(1.60590438368216133e-10);               // 1/13!
    p = fma(p, r, sgpr_const(2.08767569878681002e-09));   // 1/12!
    p = fma(p, r, sgpr_const(2.50521083854417202e-08));   // 1/11!
    p = fma(p, r, sgpr_const(2.75573192239858883e-07));   // 1/10!
    p = fma(p, r, sgpr_const(2.75573192239858925e-06));   // 1/9!
    p = fma(p, r, sgpr_const(2.48015873015873016e-05));   // 1/8!
    p = fma(p, r, sgpr_const(1.98412698412698413e-04));   // 1/7!
    p = fma(p, r, sgpr_const(1.38888888888888894e-03));   // 1/6!
    p = fma(p, r, sgpr_const(8.33333333333333322e-03));   // 1/5!
    p = fma(p, r, sgpr_const(4.16666666666666644e-02));   // 1/4!
    p = fma(p, r, sgpr_const(1.66666666666666657e-01));   // 1/3!
    p = fma(p, r, sgpr_const(5.00000000000000000e-01));   // 1/2!
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    // (Estrin's scheme -- the same polynomial in 16 instructions of depth 5 -- was measured in round 5: 643 against 647 k
    //  registrations/s on the bench; the chain through the exponential is not what the gradient term waits for)
    const double v = ldexp(p, (int)fmax(kf, -1100.0));
    return x < -745.2 ? 0.0 : v;
}

// 1 / x for a finite x of ordinary magnitude (|det(CSum)| > 1e-12): hardware reciprocal + two Newton steps -- five
// instructions against the eleven of the IEEE division sequence, last-bit accurate away from the denormal range.
NDT_D double rcp_nr(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// One (source cell, target cell) term of NDTMatcherD2D::derivativesNDT + updateGradientHessianLocal
// (SURVEY.md App. A.4).  m, C: source mean / covariance already in the target frame.
// acc: [0] score, [1..6] gradient, [7..27] upper triangle of the Hessian (row-major).
// With x = m - mu, B = (C + Cj)^-1, q = (B x, (m - C B x) x B x) [so that x^T B (dx/dp_a) ... = 2 q_a], s = -d1 exp(-d2/2 x^T B x):
//   score += s,   gradient_a += 2 f q_a,   Hessian_ab += 2 f (h_ab - d2 q_a q_b),   f = -(d2 / 2) s,
// h = half the second-order bracket of the reference formula (the factor 2 of every term is applied once, in 2 f).
// a (e_K x v) and a . (e_K x v) without the component of e_K x v that is zero by construction: written with the vectors
// {0, -v.z, v.y} etc. every such product is an instruction (x * 0 cannot be folded: it is NaN for an infinite x), 42 of the
// 385 of a pair term with its Hessian.  The terms that are left come in the order they had: for finite operands the same
// bits (a leading +-0 does not change the sum that follows it).
template <int K>
NDT_D d3 mul_ecross(const sym3 &a, d3 v)
{
    if constexpr (K == 0) return {a.xy * (-v.z) + a.xz * v.y, a.yy * (-v.z) + a.yz * v.y, a.yz * (-v.z) + a.zz * v.y};
    else if constexpr (K == 1) return {a.xx * v.z + a.xz * (-v.x), a.xy * v.z + a.yz * (-v.x), a.xz * v.z + a.zz * (-v.x)};
    else return {a.xx * (-v.y) + a.xy * v.x, a.xy * (-v.y) + a.yy * v.x, a.xz * (-v.y) + a.yz * v.x};
}
template <int K>
NDT_D double dot_ecross(d3 a, d3 v)
{
    if constexpr (K == 0) return a.y * (-v.z) + a.z * v.y;
    else if constexpr (K == 1) return a.x * v.z + a.z * (-v.x);
    else return a.x * (-v.y) + a.y * v.x;
}

// PLANAR: the term of NDTMatcherD2D_2D ({x, y, yaw}, dof_mask 0x23): of the gradient the entries 0, 1, 5, of the Hessian the six
// entries among them -- the expressions of the full term for those entries (the same bits), nothing else (the other sums stay 0;
// newton_mask decouples their dofs anyway).  One rotational vector d_2 instead of three: ~135 instead of 276 instructions per
// 64 terms with the Hessian.  Only where nobody reads the inactive entries: not under the Tikhonov term (H^T H mixes them in).
template <bool WITH_H, bool PLANAR = false>
NDT_D void pair_term(d3 m, sym3 C, d3 mu, sym3 Cj, double lfd1, double lfd2, double *acc)
{
    const d3 x = m - mu;
    const sym3 S = C + Cj;
    // CSum.computeInverseAndDetWithCheck: adjugate / determinant, |det| > 1e-12
    sym3 A;
    A.xx = S.yy * S.zz - S.yz * S.yz;
    A.xy = S.yz * S.xz - S.xy * S.zz;
    A.xz = S.xy * S.yz - S.yy * S.xz;
    const double det = S.xx * A.xx + S.xy * A.xy + S.xz * A.xz;
    if (!(fabs(det) > NDT_DET_EPS)) return;
    A.yy = S.xx * S.zz - S.xz * S.xz;
    A.yz = S.xy * S.xz - S.xx * S.yz;
    A.zz = S.xx * S.yy - S.xy * S.xy;
    const double id = rcp_nr(det);
    const d3 xB = id * mul(A, x);                    // B x
    const double l = dot(x, xB);
    if (!(l * 0.0 == 0.0)) return;                   // if(l*0 != 0) continue;
#ifdef NDT_EXP_NO_EXP
    const double sh = -lfd1 * (1.0 - lfd2 * l * 0.5);
#else
    const double sh = -lfd1 * exp_nonpos(-lfd2 * l * 0.5);
#endif
    const double f2 = -lfd2 * sh;                    // 2 f
    const d3 w = mul(C, xB);
    if constexpr (PLANAR) {
        const d3 mw = m - w;
        const double q5 = mw.x * xB.y - mw.y * xB.x;         // (cross(m - w, B x)).z
        acc[0] += sh;
        acc[1] += f2 * xB.x; acc[2] += f2 * xB.y; acc[6] += f2 * q5;
        if (!WITH_H) return;
        const sym3 B = {A.xx * id, A.xy * id, A.xz * id, A.yy * id, A.yz * id, A.zz * id};
        const d3 c2 = mul_ecross<2>(C, xB);
        const d3 r2 = d3{-w.y - c2.x, w.x - c2.y, 0.0 - c2.z};
        const d3 d2 = d3{c2.x - mw.y, c2.y + mw.x, c2.z};
        const d3 Bd2 = mul(B, d2);
        const double xBH22 = -(xB.x * m.x + xB.y * m.y);
        const double qk0 = lfd2 * xB.x, qk1 = lfd2 * xB.y, qk5 = lfd2 * q5;
        const double h55 = dot(d2, Bd2) + xBH22 + dot_ecross<2>(r2, xB);
        acc[7] += f2 * (B.xx - qk0 * xB.x);      // (0, 0)
        acc[8] += f2 * (B.xy - qk0 * xB.y);      // (0, 1)
        acc[12] += f2 * (Bd2.x - qk0 * q5);      // (0, 5)
        acc[13] += f2 * (B.yy - qk1 * xB.y);     // (1, 1)
        acc[17] += f2 * (Bd2.y - qk1 * q5);      // (1, 5)
        acc[27] += f2 * (h55 - qk5 * q5);        // (5, 5)
        return;
    }
    const d3 qr = cross(m - w, xB);                  // x^T B j_k - x^T B Z_k B x / 2,  j_k = e_k x m
    const double q[6] = {xB.x, xB.y, xB.z, qr.x, qr.y, qr.z};
    acc[0] += sh;
#pragma unroll
    for (int a = 0; a < 6; a++) acc[1 + a] += f2 * q[a];
    if (!WITH_H) return;

    const sym3 B = {A.xx * id, A.xy * id, A.xz * id, A.yy * id, A.yz * id, A.zz * id};
    // j_k = e_k x m, p_k = e_k x (B x), e_k x w: never formed (see mul_ecross).
    // With r_k = Z_k (B x) = e_k x w - C (e_k x B x) and d_k = j_k - r_k, B symmetric:
    //   j_i^T B j_k - (B r_i)^T j_k - (B r_k)^T j_i + (B r_i)^T r_k = d_i^T B d_k      (rotation x rotation block)
    //   B j_k - B r_k = B d_k                                                          (translation x rotation block)
    // -- three matrix-vector products B d_k instead of six (B j_k, B r_k) and one dot per entry instead of four (round 5:
    // 345 -> 276 vector instructions per 64 terms; the sums associate differently: rounding-level changes of the Hessian).
    d3 r[3], d[3], Bd[3];
    {
        const d3 c0 = mul_ecross<0>(C, xB), c1 = mul_ecross<1>(C, xB), c2 = mul_ecross<2>(C, xB);
        const d3 mw = m - w;
        r[0] = d3{0.0 - c0.x, -w.z - c0.y, w.y - c0.z};
        r[1] = d3{w.z - c1.x, 0.0 - c1.y, -w.x - c1.z};
        r[2] = d3{-w.y - c2.x, w.x - c2.y, 0.0 - c2.z};
        d[0] = d3{c0.x, c0.y - mw.z, c0.z + mw.y};                // e_k x (m - w) + C (e_k x B x)
        d[1] = d3{c1.x + mw.z, c1.y, c1.z - mw.x};
        d[2] = d3{c2.x - mw.y, c2.y + mw.x, c2.z};
    }
#pragma unroll
    for (int k = 0; k < 3; k++) Bd[k] = mul(B, d[k]);
    const double Bm[3][3] = {{B.xx, B.xy, B.xz}, {B.xy, B.yy, B.yz}, {B.xz, B.yz, B.zz}};
    const double Bdv[3][3] = {{Bd[0].x, Bd[0].y, Bd[0].z}, {Bd[1].x, Bd[1].y, Bd[1].z}, {Bd[2].x, Bd[2].y, Bd[2].z}};
    // x^T B H_ik, H_ik = e_i x (e_k x m), i <= k
    const double xBH[3][3] = {{-(xB.y * m.y + xB.z * m.z), xB.y * m.x, xB.z * m.x},
                              {0.0, -(xB.x * m.x + xB.z * m.z), xB.z * m.y},
                              {0.0, 0.0, -(xB.x * m.x + xB.y * m.y)}};
    double qk[6];
#pragma unroll
    for (int a = 0; a < 6; a++) qk[a] = lfd2 * q[a];
    int o = 7;
#pragma unroll
    for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int b = a; b < 6; b++) {
            double h;
            if (b < 3) {
                h = Bm[a][b];
            } else if (a < 3) {
                h = Bdv[b - 3][a];
            } else {
                const int i = a - 3, k = b - 3;
                // d_i^T B d_k + x^T B H_ik + p_i . r_k,  p_i = e_i x (B x)
                const double pr = i == 0 ? dot_ecross<0>(r[k], xB) : i == 1 ? dot_ecross<1>(r[k], xB) : dot_ecross<2>(r[k], xB);
                h = dot(d[i], Bd[k]) + xBH[i][k] + pr;
            }
            acc[o++] += f2 * (h - qk[a] * q[b]);
        }
    }
}

// LDS of one workgroup's evaluation machinery (NW waves)
struct HitCache {                   // what a wave remembers of its last evaluation (one group of source cells per wave)
    unsigned key;                   // the caller's key of the registration (0: nothing cached)
    int base;                       // first source cell of the group
    unsigned count;                 // hits in the list
    unsigned pad;
};
template <int NW>
struct EvalShared {
    // The source cells are dealt to NDT_VW = 8 SHARES (cell i -> share i mod 8) whatever the workgroup's width: a wave
    // of a wide (8-wave) workgroup sums one share, a wave of a narrow (4-wave) workgroup two, one after the other.
    // Shares, their hit lists and their partial sums are the same either way, so the two forms agree bit for bit.
    // A share remembers the hit list of one group of 64 cells (registrations of up to 512 source cells are covered).
    static constexpr int QL = NW >= 8 ? 2048 : 1024;   // entries of one hit list (source lane << 24 | target cell)
    double src[NW * 9 * 64];        // per wave: the transformed source tile, one column per lane
    uint32_t queue[NDT_VW * QL];    // per share: the hit list
    uint2 win[NW * 7 * 64];         // per wave: decoded probe windows (up to 7 runs x 64 lanes)
    int cell[NDT_VW * 3 * 64];      // per share: the target-grid index of every lane's transformed mean, last evaluation
    HitCache cache[NDT_VW];
    double part[NDT_VW * 32];       // the shares' partial sums
    double sums[32];                // the evaluation's result
};

// What a wave carries through an evaluation: its accumulators, its hit queue, its share of the LDS.
#if defined(NDT_MATCH_PROF) && !defined(NDT_MATCH_TL)
#define NDT_MATCH_TL
#endif
#ifdef NDT_MATCH_TL
// timeline of one launch of the persistent matcher (100 MHz wall clock): per pair {first start, parked, resumed, end},
// then per workgroup the time its last wave left, then [.. + 1024] the launch's first time stamp
#define NDT_TL_PAIRS 4096
__device__ long long g_tl[4 * NDT_TL_PAIRS + 1024 + 8];
#define NDT_TL(pair, k) { if ((pair) < NDT_TL_PAIRS) g_tl[4 * (pair) + (k)] = (long long)wall_clock64(); }
#else
#define NDT_TL(pair, k)
#endif
#ifdef NDT_MATCH_PROF   // experiments: section clocks of wave 0 (src+transform, probe, pop, term, reduce)
__device__ long long g_prof[16];   // [0..5]: gradient-only evaluations (5 sections + count), [8..13]: with Hessian
__device__ long long g_wave[16];   // eval_derivs: per wave, clocks until it reaches the final barrier [0..7], its pair terms [8..15]
#ifndef NDT_PROF_TID
#define NDT_PROF_TID 0
#endif
#define NDT_PROF_T(k) { if ((threadIdx.x & 63u) == NDT_PROF_TID) { long long n_ = clock64(); w.prof[k] += n_ - w.pt; w.pt = n_; } }
#else
#define NDT_PROF_T(k)
#endif
template <bool WITH_H>
struct WaveEval {
    static constexpr int NACC = WITH_H ? 28 : 7;
#ifdef NDT_MATCH_PROF
    long long prof[6], pt, pt0;
#endif
    double acc[NACC];
    double *mysrc;
    uint32_t *myq;
    uint2 *mywin;
    int *mycell;
    HitCache *cache;
    unsigned terms;                  // wave-uniform
};

// inclusive scan over the 64 lanes in the vector ALU: DPP row shifts within the rows of 16, then the two row broadcasts
// hand the row totals on (no LDS round trips)
NDT_D unsigned wave_incl_scan_u32(unsigned v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, true);   // row_bcast:31 into rows 2 and 3
    return (unsigned)x;
}

template <bool WITH_H>
NDT_D double wave_totals(const WaveEval<WITH_H> &w);      // (defined below, used by eval_group's segment rows)

// TERM stage: the n (source lane, target cell) pairs at the head of the wave's list, 64 at a time; every lane does one
// dense pair term.  Without a Hessian the target cells of the next batch are fetched while this one is computed.
template <bool WITH_H, bool PLANAR = false>
NDT_D void term_list(WaveEval<WITH_H> &w, const MapView &tg, double lfd1, double lfd2, unsigned n)
{
    const unsigned lane = threadIdx.x & 63u;
    auto tile_m = [&](unsigned sl) { return d3{w.mysrc[0 * 64 + sl], w.mysrc[1 * 64 + sl], w.mysrc[2 * 64 + sl]}; };
    auto tile_C = [&](unsigned sl) {
        return sym3{w.mysrc[3 * 64 + sl], w.mysrc[4 * 64 + sl], w.mysrc[5 * 64 + sl],
                    w.mysrc[6 * 64 + sl], w.mysrc[7 * 64 + sl], w.mysrc[8 * 64 + sl]};
    };
    // The list entries and target cells of the NEXT batch are fetched while this one is computed.  Every lane fetches
    // (lanes past the end of the list fetch its last entry again): a fetch under a lane mask is a branch the wave may
    // skip, and the wait counts the compiler then has to use (the smaller of the two paths) would make every batch wait
    // for the prefetch it has just issued.
    auto fetch = [&](unsigned e, uint32_t &en, d3 &mu, sym3 &Cj) {
        en = w.myq[min(e + lane, n - 1u)];
#ifdef NDT_EXP_FIXED_TARGET
        gcell_ptr tc = tg.cells + (en & 0x1u);
#else
        gcell_ptr tc = tg.cells + (en & 0xFFFFFFu);
#endif
        mu = d3{tc->mean[0], tc->mean[1], tc->mean[2]};
        Cj = sym3{tc->cov[0], tc->cov[1], tc->cov[2], tc->cov[3], tc->cov[4], tc->cov[5]};
    };
    if (n == 0u) return;
    if constexpr (WITH_H) {
        uint32_t en0, en1;
        d3 mu0, mu1;
        sym3 Cj0, Cj1;
        fetch(0u, en0, mu0, Cj0);
#pragma unroll 1
        for (unsigned e0 = 0; e0 < n; e0 += 64u) {
            fetch(e0 + 64u, en1, mu1, Cj1);
            __builtin_amdgcn_sched_barrier(0);       // (the loads stay above the arithmetic)
            if (e0 + lane < n) pair_term<true, PLANAR>(tile_m(en0 >> 24), tile_C(en0 >> 24), mu0, Cj0, lfd1, lfd2, w.acc);
            en0 = en1; mu0 = mu1; Cj0 = Cj1;
        }
    } else {
        // two register sets that take turns (no rotation moves: they would be a tenth of the gradient-only term)
        uint32_t enA, enB;
        d3 muA, muB;
        sym3 CjA, CjB;
        fetch(0u, enA, muA, CjA);
#pragma unroll 1
        for (unsigned e0 = 0; e0 < n; e0 += 128u) {
            fetch(e0 + 64u, enB, muB, CjB);
            __builtin_amdgcn_sched_barrier(0);
            if (e0 + lane < n) pair_term<false, PLANAR>(tile_m(enA >> 24), tile_C(enA >> 24), muA, CjA, lfd1, lfd2, w.acc);
            fetch(e0 + 128u, enA, muA, CjA);
            __builtin_amdgcn_sched_barrier(0);
            if (e0 + 64u + lane < n) pair_term<false, PLANAR>(tile_m(enB >> 24), tile_C(enB >> 24), muB, CjB, lfd1, lfd2, w.acc);
        }
    }
    w.terms += n;
}

// One group of up to 64 source cells base + k stride < end: transform (pseudoTransformNDT), PROBE, TERM.
//   PROBE reads the (2n+1)^3 slots around every lane's cell as bit windows of the map's rank bitmap (1 bit per slot +
//   the rank of every 32-slot word's first Gaussian cell): slots are z-fastest, so the neighbours along z are one run
//   of <= W bits, and in a flat map (sz <= n+1: every z-layer is a neighbour) the whole (y, z) block of one x is a
//   single run of <= W*sz bits.  Cells are ranked in slot order, so the k-th set bit of a run is cell `first + k`: no
//   per-slot lookups.  Every lane counts its hits, a scan over the wave gives it its place in the wave's hit list, and
//   the lanes fill the list without talking to each other.
//   The list only depends on the target-grid cells the transformed means fall into.  `cache_key` != 0 names the
//   registration this evaluation belongs to: when the wave's last evaluation had the same key and group and no mean
//   has left its cell since (the small steps of a line search), PROBE is skipped and the list is used again.
// On return every hit of the group has been summed into w.acc.
// SEG: the 64 lanes (cells) of the group are cut into segments of `seg_lanes` lanes -- the chunks of the grid-barrier
// matcher -- and every segment gets its own hit list, its own pass over the pair terms and its own row of wave totals
// (seg_rows + segment * seg_stride; the accumulators are zero again afterwards): TRANSFORM and PROBE run once for all 64
// cells, while the sums of a segment are bit for bit those of a group that holds that segment's cells alone.
template <int NN, bool WITH_H, int QL, bool SEG = false, bool KEEP_RUNS = false, bool PLANAR = false>
NDT_D void eval_group(WaveEval<WITH_H> &w, const MapView &tg, gcell_ptr src, int base, int stride, int end,
                      const rigid &T, double lfd1, double lfd2, unsigned cache_key, unsigned seg_lanes = 64u,
                      double *seg_rows = nullptr, unsigned seg_stride = 0u)
{
    constexpr int W = 2 * NN + 1;
    constexpr int NDT_QL = QL;
    const unsigned lane = threadIdx.x & 63u;
    const int i = base + (int)lane * stride;
    const bool vi = i < end;
    int ix = 0, iy = 0, iz = 0;
    NDT_PROF_T(4)
    // (wave-uniform: a branch, not a select.  Only in the persistent matcher of the 2D batches, whose map views live in LDS: the
    //  large-map kernels -- KEEP_RUNS -- hold theirs in scalar registers and have none to spare for the reciprocal)
    const bool res_pow2 = !KEEP_RUNS && __builtin_amdgcn_readfirstlane(tg.inv_res != 0.0 ? 1 : 0) != 0;
    if (vi) {
        gcell_ptr sc = src + i;
        d3 m0 = {sc->mean[0], sc->mean[1], sc->mean[2]};
        sym3 C0 = {sc->cov[0], sc->cov[1], sc->cov[2], sc->cov[3], sc->cov[4], sc->cov[5]};
        d3 m = apply(T, m0);                    // pseudoTransformNDT: mean' = T mean
        sym3 C = rotate_cov(T.r, C0);           //                      cov'  = R cov R^T
        w.mysrc[0 * 64 + lane] = m.x; w.mysrc[1 * 64 + lane] = m.y; w.mysrc[2 * 64 + lane] = m.z;
        w.mysrc[3 * 64 + lane] = C.xx; w.mysrc[4 * 64 + lane] = C.xy; w.mysrc[5 * 64 + lane] = C.xz;
        w.mysrc[6 * 64 + lane] = C.yy; w.mysrc[7 * 64 + lane] = C.yz; w.mysrc[8 * 64 + lane] = C.zz;
        if (res_pow2) {                                     // getCellsForPoint(mean, n_neighbours)
            ix = lazygrid_index_p2(m.x, tg.cx, tg.inv_res, tg.hx);
            iy = lazygrid_index_p2(m.y, tg.cy, tg.inv_res, tg.hy);
            iz = lazygrid_index_p2(m.z, tg.cz, tg.inv_res, tg.hz);
        } else {
            ix = lazygrid_index_h(m.x, tg.cx, tg.res, tg.hx);
            iy = lazygrid_index_h(m.y, tg.cy, tg.res, tg.hy);
            iz = lazygrid_index_h(m.z, tg.cz, tg.res, tg.hz);
        }
    }
    NDT_PROF_T(0)
    const HitCache hc = *w.cache;
    const bool moved = vi && (w.mycell[lane] != ix || w.mycell[64 + lane] != iy || w.mycell[128 + lane] != iz);
    const bool reuse = cache_key != 0u && hc.key == cache_key && hc.base == base && !ndt_ballot(moved);
    if (vi && !reuse) { w.mycell[lane] = ix; w.mycell[64 + lane] = iy; w.mycell[128 + lane] = iz; }
    // flat form only when every lane's z-neighbourhood is the whole column (a source cell that lies two or
    // more cells above / below a thin map sees only part of it, or nothing): wave-uniform
    const bool flat = tg.sz <= NN + 1 && !ndt_ballot(vi && !(iz - NN <= 0 && iz + NN >= tg.sz - 1));
    const int zlo = flat ? 0 : max(iz - NN, 0), zhi = flat ? tg.sz - 1 : min(iz + NN, tg.sz - 1);
    grank_ptr rmw = tg.rankmap;
    // The runs are fetched W at a time (flat map: the W runs of the W x-neighbours; otherwise, per x, the W runs of the
    // y-neighbours): all 2 W loads are in flight together.
    auto windows = [&](int outer, unsigned (&bits)[W], unsigned (&id0)[W]) {
        uint2 wa[W], wb[W];
        unsigned sh[W];
        int len[W];
#pragma unroll
        for (int q = 0; q < W; q++) {
            const int dx = flat ? q - NN : outer - NN;
            const int xx = ix + dx;
            const bool xok = vi && xx >= 0 && xx < tg.sx && zlo <= zhi;
            const int ylo = flat ? max(iy - NN, 0) : iy - NN + q;
            const int yhi = flat ? min(iy + NN, tg.sy - 1) : ylo;
            const bool ok = xok && ylo <= yhi && ylo >= 0 && yhi < tg.sy;
            const unsigned s0 = ok ? (unsigned)((xx * tg.sy + ylo) * tg.sz + zlo) : 0u;
            len[q] = ok ? (flat ? (yhi - ylo + 1) * tg.sz : zhi - zlo + 1) : 0;   // <= 28 bits
            sh[q] = s0 & 31u;
            const unsigned long long ra = rmw[s0 >> 5], rb = rmw[(s0 >> 5) + 1u];
            wa[q] = make_uint2((unsigned)ra, (unsigned)(ra >> 32));
            wb[q] = make_uint2((unsigned)rb, (unsigned)(rb >> 32));
        }
#pragma unroll
        for (int q = 0; q < W; q++) {
            bits[q] = __builtin_amdgcn_alignbit(wb[q].x, wa[q].x, sh[q]) & ((1u << len[q]) - 1u);
            const unsigned lowa = wa[q].x >> sh[q];          // the window's part of the first word
            id0[q] = lowa ? wa[q].y + (unsigned)__popc(wa[q].x & ((1u << sh[q]) - 1u)) : wb[q].y;
        }
    };
    uint2 *win = w.mywin + lane;
    unsigned cnt = 0;
    // thick maps (W x W runs of <= W bits per lane): the decoded runs stay in REGISTERS, (bits << 24 | first cell rank), for
    // the pass that fills the hit list -- fetching and decoding them a second time there was a third of a 3D evaluation
    // (16 k of 48 k clocks per 512 cells, tools/evalloop3d.py).  The largest neighbourhood (7 x 7 runs) fetches twice.
    // (KEEP_RUNS: the kernels that serve large maps; the persistent matcher of the 2D batches has no registers to spare)
    constexpr bool KEEP = KEEP_RUNS && W <= 5;
    unsigned pk[KEEP ? W * W : 1];
#pragma unroll
    for (int i = 0; i < (KEEP ? W * W : 1); i++) pk[i] = 0u;
    if (!reuse) {
        if (flat) {
            unsigned bits[W], id0[W];
            windows(0, bits, id0);
#pragma unroll
            for (int q = 0; q < W; q++) {
                win[q * 64] = make_uint2(bits[q], id0[q]);
                cnt += (unsigned)__popc(bits[q]);
            }
        } else if constexpr (KEEP) {
            // two rounds of 2 W loads in flight: round k + 1 is issued before round k is decoded (all W rounds at once would
            // need 100 registers; one at a time pays W dependent round trips)
            uint2 wa[2][W], wb[2][W];
            unsigned shf[2][W];
            int len[2][W];
            auto issue = [&](int outer, int slot) __attribute__((always_inline)) {
                const int xx = ix + outer - NN;
                const bool xok = vi && xx >= 0 && xx < tg.sx && zlo <= zhi;
#pragma unroll
                for (int q = 0; q < W; q++) {
                    const int yy = iy - NN + q;
                    const bool ok = xok && yy >= 0 && yy < tg.sy;
                    const unsigned s0 = ok ? (unsigned)((xx * tg.sy + yy) * tg.sz + zlo) : 0u;
                    len[slot][q] = ok ? zhi - zlo + 1 : 0;
                    shf[slot][q] = s0 & 31u;
                    const unsigned long long ra = rmw[s0 >> 5], rb = rmw[(s0 >> 5) + 1u];
                    wa[slot][q] = make_uint2((unsigned)ra, (unsigned)(ra >> 32));
                    wb[slot][q] = make_uint2((unsigned)rb, (unsigned)(rb >> 32));
                }
            };
            issue(0, 0);
#pragma unroll
            for (int outer = 0; outer < W; outer++) {
                const int cur = outer & 1;
                if (outer + 1 < W) issue(outer + 1, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < W; q++) {
                    const unsigned bits = __builtin_amdgcn_alignbit(wb[cur][q].x, wa[cur][q].x, shf[cur][q]) & ((1u << len[cur][q]) - 1u);
                    const unsigned lowa = wa[cur][q].x >> shf[cur][q];          // the window's part of the first word
                    const unsigned id0 = lowa ? wa[cur][q].y + (unsigned)__popc(wa[cur][q].x & ((1u << shf[cur][q]) - 1u)) : wb[cur][q].y;
                    pk[outer * W + q] = (bits << 24) | (id0 & 0xFFFFFFu);
                    cnt += (unsigned)__popc(bits);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll 1
            for (int outer = 0; outer < W; outer++) {
                unsigned bits[W], id0[W];
                windows(outer, bits, id0);
#pragma unroll
                for (int q = 0; q < W; q++) cnt += (unsigned)__popc(bits[q]);
            }
        }
    }
    // entries [p0, p1) of a hit list in which this lane's `cnt_l` hits start at `off_l` go to the queue.  FROM_REGS: the
    // runs of a thick map come from the registers the count pass left them in (only the FIRST fill of a group may: the
    // registers must not live across the pair terms); otherwise they are fetched and decoded again.
    auto put_run = [&](unsigned b_, unsigned id, unsigned &gi, unsigned p0, unsigned p1) __attribute__((always_inline)) {
        while (b_) {
            if (gi >= p0 && gi < p1) w.myq[gi - p0] = (lane << 24) | id;
            gi += 1u; id += 1u; b_ &= b_ - 1u;
        }
    };
    auto fill_flat = [&](unsigned &gi, unsigned p0, unsigned p1) __attribute__((always_inline)) {
#pragma unroll 1
        for (int q = 0; q < W; q++) {
            const uint2 v = win[q * 64];
            put_run(v.x, v.y, gi, p0, p1);
        }
    };
    auto fill_regs = [&](unsigned off_l, unsigned cnt_l, unsigned p0, unsigned p1) __attribute__((always_inline)) {
        if (!reuse && cnt_l != 0u && off_l < p1 && off_l + cnt_l > p0) {
            unsigned gi = off_l;
            if (flat) {
                fill_flat(gi, p0, p1);
            } else if constexpr (KEEP) {
                // branch-free: every bit of every run stores -- a set bit into the list, a clear one into a word of the
                // (idle, on thick maps) window buffer -- and advances the list by its value.  The loops over the set bits of
                // a run cost a wave the longest run of its 64 lanes per run plus a branch per step: 10 k clocks per 512 cells.
                // (only for the one-pass fill: p0 = 0 and the whole list fits)
                uint32_t *const dummy = reinterpret_cast<uint32_t *>(w.mywin) + lane;
                uint32_t *q = w.myq + gi;
#pragma unroll
                for (int i = 0; i < W * W; i++) {
                    const unsigned b_ = pk[i] >> 24;
                    unsigned id = (lane << 24) | (pk[i] & 0xFFFFFFu);
#pragma unroll
                    for (int b = 0; b < W; b++) {
                        const unsigned on = (b_ >> b) & 1u;
                        *(on ? q : dummy) = id;
                        q += on; id += on;
                    }
                }
            } else {                                      // (no registers kept: fetch and decode again)
#pragma unroll 1
                for (int outer = 0; outer < W; outer++) {
                    unsigned bits[W], id0[W];
                    windows(outer, bits, id0);
#pragma unroll
                    for (int q = 0; q < W; q++) put_run(bits[q], id0[q], gi, p0, p1);
                }
            }
        }
    };
    auto fill = [&](unsigned off_l, unsigned cnt_l, unsigned p0, unsigned p1) __attribute__((always_inline)) {
        if (!reuse && cnt_l != 0u && off_l < p1 && off_l + cnt_l > p0) {
            unsigned gi = off_l;
            if (flat) {
                fill_flat(gi, p0, p1);
            } else {
#pragma unroll 1
                for (int outer = 0; outer < W; outer++) {
                    unsigned bits[W], id0[W];
                    windows(outer, bits, id0);
#pragma unroll
                    for (int q = 0; q < W; q++) put_run(bits[q], id0[q], gi, p0, p1);
                }
            }
        }
    };
    // the list is materialised NDT_QL hits at a time (one pass unless the neighbourhoods are dense 3D ones), each
    // pass followed by its pair terms
    // (`filled`: the list already holds its first -- then only -- pass)
    auto run_list = [&](unsigned off_l, unsigned cnt_l, unsigned total_l, bool filled) __attribute__((always_inline)) {
#pragma unroll 1
        for (unsigned p0 = 0; p0 < total_l; p0 += (unsigned)NDT_QL) {
            const unsigned p1 = min(total_l, p0 + (unsigned)NDT_QL);
            if (!(filled && p0 == 0u)) fill(off_l, cnt_l, p0, p1);
            ndt_wave_sync();                             // list entries and tile columns were written by other lanes
            NDT_PROF_T(2)
            term_list<WITH_H, PLANAR>(w, tg, lfd1, lfd2, p1 - p0);
            NDT_PROF_T(3)
            ndt_wave_sync();                             // the next pass (or group) overwrites the list and the tile
        }
    };
    unsigned total = hc.count, my_off = 0u;
    if constexpr (SEG) {
        constexpr int NACC = WaveEval<WITH_H>::NACC, SH = WITH_H ? 1 : 3;
        NDT_PROF_T(1)
        const unsigned incl = wave_incl_scan_u32(cnt);
        const unsigned total_all = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
        const unsigned off_all = incl - cnt;
        const unsigned long long have = ndt_ballot(vi);
        const unsigned n_seg = have ? (63u - (unsigned)__builtin_clzll(have)) / seg_lanes + 1u : 0u;   // segments that hold cells
        // list offset at which segment x starts (x == n_seg: the end of the list)
        auto seg_off = [&](unsigned x) -> unsigned {
            return x * seg_lanes < 64u && x < n_seg ? (unsigned)__builtin_amdgcn_readlane((int)off_all, (int)(x * seg_lanes)) : total_all;
        };
        auto emit_row = [&](unsigned sg) __attribute__((always_inline)) {
            const double tot = wave_totals<WITH_H>(w);
            double *row = seg_rows + sg * seg_stride;
            if ((lane & ((1u << SH) - 1u)) == 0u && (lane >> SH) < (unsigned)NACC) row[lane >> SH] = tot;
            if (lane == 0) row[28] = (double)w.terms;
            w.terms = 0;
#pragma unroll
            for (int k = 0; k < NACC; k++) w.acc[k] = 0.0;
        };
        // the terms of segments [sg, e), whose hits are in the list from offset o0 on: every segment on its own
        auto seg_terms = [&](unsigned sg, unsigned e, unsigned o0) __attribute__((always_inline)) {
            uint32_t *const q0 = w.myq;
#pragma unroll 1
            for (unsigned x = sg; x < e; x++) {
                const unsigned a = seg_off(x), b = seg_off(x + 1u);
                w.myq = q0 + (a - o0);
                term_list<WITH_H>(w, tg, lfd1, lfd2, b - a);
                emit_row(x);
            }
            w.myq = q0;
        };
        // the usual case: ONE fill, from the registers, for all segments -- before the loop, so that the registers are dead
        // when the pair terms start
        const bool all_in = total_all <= (unsigned)NDT_QL;
        if (all_in) fill_regs(off_all, cnt, 0u, total_all);
#pragma unroll 1
        for (unsigned sg = 0; sg < n_seg;) {
            const unsigned o0 = seg_off(sg);
            unsigned e = sg + 1u;
            if (all_in) e = n_seg;
            else while (e < n_seg && seg_off(e + 1u) - o0 <= (unsigned)NDT_QL) e++;   // whole segments that fit into the list together
            const unsigned o1 = seg_off(e);
            const bool in = lane >= sg * seg_lanes && lane < e * seg_lanes;
            const unsigned cnt_s = in ? cnt : 0u, off_s = in ? off_all - o0 : 0u;
            if (o1 - o0 > (unsigned)NDT_QL) {             // one segment with more hits than the list holds: its own passes
                run_list(off_s, cnt_s, o1 - o0, false);
                emit_row(sg);
            } else {                                      // one pass of window fetches for all of them, then their terms in turn
                if (!all_in) fill(off_s, cnt_s, 0u, o1 - o0);
                ndt_wave_sync();
                seg_terms(sg, e, o0);
                ndt_wave_sync();
            }
            sg = e;
        }
        return;
    }
    if (!reuse) {
        const unsigned incl = wave_incl_scan_u32(cnt);
        total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
        my_off = incl - cnt;
    }
    NDT_PROF_T(1)
    const bool one_pass = total <= (unsigned)NDT_QL;      // the usual case: one fill, from the registers (dead after it)
    if (one_pass) fill_regs(my_off, cnt, 0u, total);
    run_list(my_off, cnt, total, one_pass);
    if (lane == 0 && !reuse) {
        HitCache nc;
        nc.key = total <= (unsigned)NDT_QL ? cache_key : 0u;
        nc.base = base; nc.count = total; nc.pad = 0u;
        *w.cache = nc;
    }
}

template <int NW, bool WITH_H>
NDT_D void wave_eval_init(WaveEval<WITH_H> &w, EvalShared<NW> &sh)
{
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    w.mysrc = sh.src + wave * (9 * 64);
    w.mywin = sh.win + wave * (7 * 64);
    w.myq = sh.queue; w.mycell = sh.cell; w.cache = sh.cache;     // (set per share)
    w.terms = 0;
#ifdef NDT_MATCH_PROF
    for (int k = 0; k < 6; k++) w.prof[k] = 0;
    w.pt = clock64(); w.pt0 = w.pt;
#endif
}

// wave-wide totals of the accumulators: on return lane l holds value (l >> SH) in `tot` when (l & ((1 << SH) - 1)) == 0
template <bool WITH_H>
NDT_D double wave_totals(const WaveEval<WITH_H> &w)
{
    constexpr int NACC = WaveEval<WITH_H>::NACC, NP = WITH_H ? 32 : 8;
    double vv[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) vv[k] = k < NACC ? w.acc[k] : 0.0;
    return wave_sum_all<NP>(vv);
}

// ---- range-partitioned evaluation (cooperative / host-driven / stand-alone kernels) --------------------------
// One evaluation of derivativesNDT over source cells [0, msrc), transformed by T, on the 8 waves of a wide
// workgroup: the cells are dealt to the waves in turn (even a small range keeps all waves busy).  Result in sh.sums[0..6]
// ([7..27] when WITH_H, [28] = pair terms).  Ends with a barrier.
template <int NN, bool WITH_H, int NW = NDT_MATCH_WAVES>
NDT_D void eval_derivs(const MapView &tg, gcell_ptr src, int msrc, const rigid &T, double lfd1,
                       double lfd2, EvalShared<NW> &sh, unsigned cache_key = 0u)
{
    constexpr int NACC = WaveEval<WITH_H>::NACC, SH = WITH_H ? 1 : 3, QL = EvalShared<NW>::QL;
    unsigned tid = threadIdx.x;
    asm volatile("" : "+v"(tid));       // (opaque: LDS addresses derived from it stay inside this call, see eval_chunks)
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), lane = tid & 63u;
    WaveEval<WITH_H> w;
    wave_eval_init<NW, WITH_H>(w, sh);
    // cells are dealt to the shares in turn (cell i -> share i mod 8): cells are ranked in slot order, so a contiguous
    // part would be one strip of the map, and strips differ a lot in how many neighbours their cells have
    const unsigned nsh = ndt_shares_of(msrc);
    const unsigned key = msrc <= 64 * (int)nsh ? cache_key : 0u;   // a share remembers the hit list of ONE group
#pragma unroll 1
    for (unsigned v = wave; v < (unsigned)NDT_VW; v += (unsigned)NW) {
        w.myq = sh.queue + v * QL;
        w.mycell = sh.cell + v * (3 * 64);
        w.cache = sh.cache + v;
        w.terms = 0;
#pragma unroll
        for (int k = 0; k < NACC; k++) w.acc[k] = 0.0;
        if (v < nsh)
            for (int base = (int)v; base < msrc; base += 64 * (int)nsh)
                eval_group<NN, WITH_H, QL, false, true>(w, tg, src, base, (int)nsh, msrc, T, lfd1, lfd2, key);
        const double tot = wave_totals<WITH_H>(w);
        if ((lane & ((1u << SH) - 1u)) == 0u && (lane >> SH) < (unsigned)NACC) sh.part[v * 32 + (lane >> SH)] = tot;
        if (lane == 0) sh.part[v * 32 + 28] = (double)w.terms;
#ifdef NDT_MATCH_PROF
        if (lane == 0) { atomicAdd((unsigned long long *)&g_wave[v & 7u], (unsigned long long)(clock64() - w.pt0)); atomicAdd((unsigned long long *)&g_wave[8u + (v & 7u)], (unsigned long long)w.terms); }
#endif
    }
    __syncthreads();
    if (tid < (unsigned)NACC || tid == 28u) {
        double s = 0;
        for (int k = 0; k < NDT_VW; k++) s += sh.part[k * 32 + tid];
        sh.sums[tid] = s;             // [28]: number of (source, target) pair terms of this evaluation
    }
    __syncthreads();
    NDT_PROF_T(4)
#ifdef NDT_MATCH_PROF
    if (tid == NDT_PROF_TID) { for (int k = 0; k < 5; k++) atomicAdd((unsigned long long *)&g_prof[k + (WITH_H ? 8 : 0)], (unsigned long long)w.prof[k]); atomicAdd((unsigned long long *)&g_prof[5 + (WITH_H ? 8 : 0)], 1ull); }
#endif
}

// The same for up to 64 / seg_lanes CHUNKS of 8 * seg_lanes source cells at once (msrc <= 512 cells, one group of 64
// cells per share, a chunk = seg_lanes lanes of it): TRANSFORM and PROBE use all 64 lanes, every chunk gets its own sums.
// Row `c` of `out` (32 doubles, as sh.sums) is bit for bit what eval_derivs yields for chunk c alone.  Ends with a barrier.
template <int NN, bool WITH_H>
NDT_D void eval_chunks(const MapView &tg, gcell_ptr src, int msrc, const rigid &T, double lfd1, double lfd2,
                       EvalShared<NDT_MATCH_WAVES> &sh, unsigned seg_lanes, double *rows, double *out, unsigned n_out)
{
    constexpr int NACC = WaveEval<WITH_H>::NACC, QL = EvalShared<NDT_MATCH_WAVES>::QL;
    unsigned tid = threadIdx.x;
    asm volatile("" : "+v"(tid));       // (opaque: LDS addresses derived from it stay inside this call instead of living, spilled, in the caller's loop)
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const unsigned n_seg = ((unsigned)msrc + 8u * seg_lanes - 1u) / (8u * seg_lanes);
    WaveEval<WITH_H> w;
    wave_eval_init<NDT_MATCH_WAVES, WITH_H>(w, sh);
    {
        const unsigned v = wave;                      // 8 waves, 8 shares
        w.myq = sh.queue + v * QL;
        w.mycell = sh.cell + v * (3 * 64);
        w.cache = sh.cache + v;
        w.terms = 0;
#pragma unroll
        for (int k = 0; k < NACC; k++) w.acc[k] = 0.0;
        // a share whose lanes hold no cell of a chunk (the tail of the range) must still deliver a row of zeros for it
        for (unsigned sg = 0; sg < n_seg; sg++)
            if ((tid & 63u) < 32u) rows[(sg * NDT_VW + v) * 32 + (tid & 63u)] = 0.0;
        ndt_wave_sync();
        eval_group<NN, WITH_H, QL, true, true>(w, tg, src, (int)v, NDT_VW, msrc, T, lfd1, lfd2, 0u, seg_lanes, rows + v * 32, NDT_VW * 32);
    }
    __syncthreads();
    for (unsigned i = tid; i < n_out * 32u; i += NDT_MATCH_THREADS) {      // (n_out >= n_seg: chunks without cells get zeros)
        const unsigned sg = i >> 5, k = i & 31u;
        if (k < (unsigned)NACC || k == 28u) {
            double a = 0;
            if (sg < n_seg)
                for (int q = 0; q < NDT_VW; q++) a += rows[(sg * NDT_VW + q) * 32 + k];
            out[sg * 32 + k] = a;
        }
    }
    __syncthreads();
}

}  // namespace

// ---- the persistent matcher: a scheduler of WAVE tasks ------------------------------------------------------------
// A workgroup (8 waves, one per CU: 256 VGPRs each) keeps R registrations in flight ("slots", all state in LDS).  The
// unit of work is a SHARE of an evaluation: the source cells of a registration are dealt to NDT_VW = 8 shares, and a
// wave that has nothing to do takes the next share of whichever slot has one -- transform, PROBE, TERM, wave sum, one
// row of partial sums into LDS.  The wave that delivers the LAST share of an evaluation adds the eight rows in fixed
// order and runs the solver step (csrc/ndt_solver.h) right there: its accumulators are dead, nothing lives across the
// step, and the other seven waves are already working on the other slot's shares.  No workgroup barrier after the
// start:
//   * a wave never waits for its workgroup's slowest wave, and the serial solver step of one registration hides
//     behind the pair terms of the other (the CU-time of a registration drops by a third);
//   * when only one registration is left on a CU its 8 shares occupy the 8 waves: a lone registration is as fast as
//     it can be on one CU.  The same kernel serves the bulk and the tail.
// Shares, their hit lists and their partial sums are the same whoever computes them and the rows are added in share
// order: a registration's result does not depend on R, on the batch it is in or on timing -- bit for bit.
//
// Work distribution across workgroups.  Registrations differ 10x in length (most converge in 5-9 Newton iterations, a
// few run into ITR_MAX with long line searches) and a long one that STARTS late sets the launch time.  Slots pull
// pairs from a ticket counter, and a registration that is still running after `park_iters` iterations is parked (its
// solver state, ~1.5 KB, goes to global memory) whenever a pair that has not started yet can be taken instead: every
// registration starts before any long tail is run.  When the tickets are gone the parked registrations are resumed, at
// most ONE of them per workgroup unless more are waiting than there are workgroups (a resumed registration is a long
// one: two of them on one CU would run at half speed each while other CUs have nothing left to do).
//   fresh   : next pair that has not started
//   reserve : parked-list slots handed out;  head : parked-list slots consumed
//   ids[s]  : 0 = slot not written yet, 1 = cancelled, pair + 2 = parked pair
// A list slot is reserved BEFORE the fresh ticket is drawn, so whoever sees the tickets exhausted and then
// head == reserve knows that no registration can be parked any more; a workgroup whose slots have all seen that exits,
// and its CU is free for the next launch (bench.py's pipeline).
struct NdtMatchWork {
    unsigned fresh, reserve, head;
    unsigned abort;        // a wave that found nothing to do for about a second raises it and everybody leaves (guard against
                           // a scheduling bug hanging the device; ndtgpu_match_aborted())
};
struct NdtParkedState {
    MatchState st;
    unsigned long long cnt[4];
};
size_t ndt_match_abort_offset() { return offsetof(NdtMatchWork, abort); }
size_t ndt_match_work_bytes(size_t n_pairs, size_t n_slots)
{
    return sizeof(NdtMatchWork) + (n_pairs + n_slots + 1) * sizeof(unsigned) + 8 + n_pairs * sizeof(NdtParkedState);
}

namespace {
enum { SLOT_FREE = 0, SLOT_BUSY = 1, SLOT_RUN = 2, SLOT_CLOSED = 3 };
enum { TASK_NONE = -1, TASK_EXIT = -2, TASK_LOAD = 0x100 };

template <int QL>
struct MatchSlot {
    MatchState st;
    NewtonWs ws;
    MapView tg, sv;
    unsigned long long cnt[4];      // wave cycles in share tasks, wave cycles in solver steps, pair terms g / h
    uint32_t queue[NDT_VW * QL];    // per share: the hit list
    int cell[NDT_VW * 3 * 64];      // per share: the target-grid index of every lane's transformed mean, last evaluation
    HitCache cache[NDT_VW];
    double part[NDT_VW * 32];       // the shares' partial sums
    double sums[32];                // the evaluation's result
    unsigned state;                 // SLOT_*
    unsigned next;                  // next share to hand out (>= NDT_VW: none)
    unsigned done;                  // shares delivered
    unsigned session;               // names the registration (key of the hit lists)
    unsigned pair;
    unsigned retry;                 // shader clock (low word) before which a FREE slot is not looked at again
    int with_h;                     // the request: evaluate with the Hessian?
    int preset;                     // a fresh ticket drawn while parking (-1: none)
    int resumed;                    // the registration came from the parked list
    // matchFusion feature / odometry-cell maps with known correspondence (fusion.h:858-871): n_feat pairs of
    // {source mean[3], cov[6], target mean[3], cov[6]} in global memory, their sums at the Newton pose / a trial pose
    const double *feat;
    unsigned n_feat;
    int code;                       // hand-over of a stage's NEXT_* code from lane 0 to the wave
    double fsums[32], ftrial[8];
    long long t_pub;                // (timeline builds) shader clock when the running request was published
    unsigned long long wall;        // (timeline builds) sum over evaluations of publish -> last share delivered
};

// one share of one evaluation: on return the wave's row of partial sums is in S.part
template <int NN, bool WITH_H, int QL, bool PLANAR = false>
NDT_D void run_share(MatchSlot<QL> &S, unsigned v, double *wsrc, uint2 *wwin, double lfd1, double lfd2)
{
    constexpr int NACC = WaveEval<WITH_H>::NACC, SH = WITH_H ? 1 : 3;
    const unsigned lane = threadIdx.x & 63u;
    WaveEval<WITH_H> w;
    w.mysrc = wsrc; w.mywin = wwin;
    w.myq = S.queue + v * QL;
    w.mycell = S.cell + v * (3 * 64);
    w.cache = S.cache + v;
    w.terms = 0;
#ifdef NDT_MATCH_PROF
    for (int k = 0; k < 6; k++) w.prof[k] = 0;
    w.pt = clock64(); w.pt0 = w.pt;
#endif
#pragma unroll
    for (int k = 0; k < NACC; k++) w.acc[k] = 0.0;
    const int msrc = S.sv.n_cells;
    const unsigned nsh = ndt_shares_of(msrc);
    if (v >= nsh) {                                              // (a share without cells: a row of zeros, no butterfly)
        if (lane < 29u) S.part[v * 32 + lane] = 0.0;
        return;
    }
    const unsigned key = msrc <= 64 * (int)nsh ? S.session : 0u;   // a share remembers the hit list of ONE group
    for (int base = (int)v; base < msrc; base += 64 * (int)nsh)
        eval_group<NN, WITH_H, QL, false, false, PLANAR>(w, S.tg, S.sv.cells, base, (int)nsh, msrc, S.st.Teval, lfd1, lfd2, key);
    const double tot = wave_totals<WITH_H>(w);
    if ((lane & ((1u << SH) - 1u)) == 0u && (lane >> SH) < (unsigned)NACC) S.part[v * 32 + (lane >> SH)] = tot;
    if (lane == 0) S.part[v * 32 + 28] = (double)w.terms;
    NDT_PROF_T(4)
#ifdef NDT_MATCH_PROF
    if (lane == 0) { for (int k = 0; k < 5; k++) atomicAdd((unsigned long long *)&g_prof[k + (WITH_H ? 8 : 0)], (unsigned long long)w.prof[k]); atomicAdd((unsigned long long *)&g_prof[5 + (WITH_H ? 8 : 0)], 1ull); }
#endif
}

template <int QL>
NDT_D void slot_result(const MatchSlot<QL> &S, double *T16, NdtMatchResultDev *res)
{
    NdtMatchResultDev &o = res[S.pair];
    match_state_result(S.st, T16 + (size_t)S.pair * 16, o);
    o.n_source = S.sv.n_cells;
    o.n_target = S.tg.n_cells;
    o.cycles_eval = (long long)(S.cnt[0] / NDT_MATCH_WAVES);    // CU-time: wave cycles / waves of a CU
    o.cycles_solver = (long long)S.cnt[1];                      // wave cycles of the solver steps
    o.pair_terms_g = (long long)S.cnt[2];
    o.pair_terms_h = (long long)S.cnt[3];
#ifdef NDT_MATCH_TL
    o.pair_terms_g = (long long)S.wall;      // timeline builds: wall clocks of the evaluations instead
#endif
}

// NDTMatcherFeatureD2D::derivativesNDT: the pair terms of the n <= 64 correspondences, source cells moved by T
// (pseudoTransformNDT), one term per lane; out[0] score, [1..6] gradient, [7..27] Hessian (WITH_H).  All lanes.
template <bool WITH_H>
NDT_D void feat_eval(const double *__restrict__ feat, unsigned n, const rigid &T, double lfd1, double lfd2, double *out)
{
    constexpr int NACC = WITH_H ? 28 : 7, NP = WITH_H ? 32 : 8, SH = WITH_H ? 1 : 3;
    const unsigned lane = threadIdx.x & 63u;
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; k++) acc[k] = 0.0;
    if (lane < n) {
        const double *c = feat + (size_t)lane * 18;
        const d3 m = apply(T, d3{c[0], c[1], c[2]});
        const sym3 C = rotate_cov(T.r, sym3{c[3], c[4], c[5], c[6], c[7], c[8]});
        pair_term<WITH_H>(m, C, d3{c[9], c[10], c[11]}, sym3{c[12], c[13], c[14], c[15], c[16], c[17]}, lfd1, lfd2, acc);
    }
    double vv[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) vv[k] = k < NACC ? acc[k] : 0.0;
    const double tot = wave_sum_all<NP>(vv);
    if ((lane & ((1u << SH) - 1u)) == 0u && (lane >> SH) < (unsigned)NACC) out[lane >> SH] = tot;
}

// matcher_feat_d2d.lineSearchMT(pose_increment_v, nextNDT_feat, targetNDT_feat) (fusion.h:1015-1017) after the NDT line
// search, then fusion.h:1018-1032: the smaller of the two steps (the larger one when either is 0) and the pose update.
// The More-Thuente driver is the one of the NDT search (same block st.mt, same stages), fed with the feature sums; a
// trial costs one pair-term batch of this wave, no share tasks.  Its evaluations are not counted in `fevals` (which
// counts derivative evaluations of the NDT maps).  All lanes.
template <int QL>
NDT_D void feat_linesearch_and_apply(MatchSlot<QL> &S, const NdtMatchParamsDev &prm)
{
    const unsigned lane = threadIdx.x & 63u;
    MatchState &st = S.st;
    if (lane == 0) {
        st.step_ndt = st.step_size;
        st.fevals_saved = st.fevals;
        st.ls_fconst = 0.0;
        for (int a = 0; a < 6; a++) st.ls_gconst[a] = 0.0;
        // what survives of the NDT search: was its first trial (evaluated with its Hessian) accepted at full length?
        S.code = (st.reuse_sums ? 4 : 0) | (st.spec_ok ? 8 : 0);
        double dg0 = 0;
        for (int a = 0; a < 6; a++) dg0 += st.incr[a] * S.fsums[1 + a];
        if (dg0 >= 0.0) S.code |= 16;                      // the feature search negates the increment in place
        S.code |= mt_start(st, S.fsums[0], S.fsums + 1) << 8;
    }
    ndt_wave_sync();
    const int keep = S.code & 0xff;
    int next = S.code >> 8;
    while (next == NEXT_REQUEST_TRIAL) {
        if (lane == 0) mt_request_trial(st);
        ndt_wave_sync();
        feat_eval<false>(S.feat, S.n_feat, st.Teval, prm.lfd1, prm.lfd2, S.ftrial);
        ndt_wave_sync();
        if (lane == 0) S.code = linesearch_step(st, S.ftrial, prm);
        ndt_wave_sync();
        next = S.code;
    }
    if (lane == 0) {
        const double step_feat = st.step_size, step_ndt = st.step_ndt;
        double step;
        if (step_ndt != 0.0 && step_feat != 0.0) step = step_ndt < step_feat ? step_ndt : step_feat;
        else step = step_ndt > step_feat ? step_ndt : step_feat;
        st.fevals = st.fevals_saved;
        st.spec_ok = (keep & 8) ? 1 : 0;
        // the Hessian sums of an accepted first NDT trial belong to the pose T would have had after that step
        st.reuse_sums = ((keep & 4) && !(keep & 16) && step == step_ndt) ? 1 : 0;
        st.step_size = step;
        apply_step(st, prm);
    }
    ndt_wave_sync();
}

// The solver step of a slot whose evaluation (S.sums) is complete.  Without feature maps: match_state_step on lane 0.
// With them the wave walks the same stages itself, because the feature sums are evaluated by all of its lanes.
// (Round 5 measured the trial / step poses on TWELVE LANES -- three lanes take the (sin, cos) pairs, twelve the entries of
//  TR = Trans Rx Ry Rz and of TR * T, operands handed over through a table in LDS, bit for bit the serial result: a lone
//  ITR_MAX registration took 2.05 instead of 1.93 ms.  What a solver stage waits for is not its arithmetic but its LDS round
//  trips -- five hand-overs through LDS cost more than the ~150 dependent instructions they replaced.  tools/timeline_match.py.)
template <int QL>
NDT_D void slot_step(MatchSlot<QL> &S, const NdtMatchParamsDev &prm)
{
    const unsigned lane = threadIdx.x & 63u;
    MatchState &st = S.st;
    if (S.n_feat == 0u) {
        if (lane == 0) match_state_step(st, S.sums, prm, S.ws);
        return;
    }
    int phase = st.phase;
    if (phase == PH_LS_TRIAL) {
        if (lane == 0) { st.reuse_sums = 0; S.code = linesearch_step(st, S.sums, prm); }
        ndt_wave_sync();
        if (S.code == NEXT_REQUEST_TRIAL) { if (lane == 0) mt_request_trial(st); return; }
        if (st.ls_joint) {                                // lineSearchMTFusion: its step is the step
            if (lane == 0) apply_step(st, prm);
            ndt_wave_sync();
        } else {
            feat_linesearch_and_apply(S, prm);            // the NDT search is over: st.step_size = step_size_ndt
        }
        if (!st.reuse_sums) return;
        phase = st.phase;
    }
    if (phase == PH_NEWTON) {
        feat_eval<true>(S.feat, S.n_feat, st.Teval, prm.lfd1, prm.lfd2, S.fsums);
        ndt_wave_sync();
        if (lane == 0) {
            if (st.ls_joint) {                            // the feature maps' share of lineSearchMTFusion's function
                st.ls_fconst = S.fsums[0];
                for (int a = 0; a < 6; a++) st.ls_gconst[a] = S.fsums[1 + a];
            }
            S.code = newton_solve(st, S.sums, S.fsums, prm, S.ws);
        }
        ndt_wave_sync();
        const int next = S.code;
        if (next == NEXT_NONE) return;
        if (next == NEXT_REQUEST_TRIAL) { if (lane == 0) mt_request_trial(st); return; }
        // NEXT_APPLY_STEP: no step control (full step), or the NDT search gave its recovery step at once
        if (!prm.step_control || st.ls_joint) { if (lane == 0) apply_step(st, prm); return; }
        if (lane == 0) st.reuse_sums = 0;
        feat_linesearch_and_apply(S, prm);
    } else if (phase == PH_FINAL) {
        feat_eval<false>(S.feat, S.n_feat, st.Teval, prm.lfd1, prm.lfd2, S.fsums);
        ndt_wave_sync();
        if (lane == 0) match_state_final(st, S.sums, S.fsums);
    }
}

NDT_D unsigned lds_load(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
NDT_D void lds_store(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
NDT_D unsigned clock_lo() { return (unsigned)__builtin_readcyclecounter(); }
}  // namespace

template <int NN, int R>
__global__ __launch_bounds__(NDT_MATCH_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void ndt_match_kernel(
    NdtSetView tset, const uint32_t *__restrict__ tidx, NdtSetView sset, const uint32_t *__restrict__ sidx,
    double *__restrict__ T16, NdtMatchParamsDev prm, NdtMatchResultDev *__restrict__ res,
    const double *__restrict__ Q36 /* per pair Tcov^-1 (matchFusion) or NULL */,
    const unsigned *__restrict__ feat_off /* [n_pairs + 1] first correspondence of every pair, or NULL */,
    const double *__restrict__ feat_cells /* 18 doubles per correspondence */,
    unsigned n_pairs, int park_iters, unsigned double_thresh, char *__restrict__ work_mem)
{
    constexpr int QL = R >= 3 ? 640 : 1024;     // (three slots: 5/8 of the hit list per share -- what fits; longer lists take two passes)
    typedef MatchSlot<QL> Slot;
    __shared__ Slot slots[R];
    __shared__ double w_src[NDT_MATCH_WAVES * 9 * 64];   // per wave: the transformed source tile, one column per lane
    __shared__ uint2 w_win[NDT_MATCH_WAVES * 7 * 64];    // per wave: decoded probe windows (up to 7 runs x 64 lanes)
    __shared__ NdtMatchParamsDev s_prm;
    __shared__ unsigned s_session, s_resumed, s_closed;

    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    if (tid == 0) { s_prm = prm; s_session = 0u; s_resumed = 0u; s_closed = 0u; }
    if (tid < (unsigned)R) {
        Slot &S = slots[tid];
        S.state = SLOT_FREE; S.next = NDT_VW; S.done = 0u; S.retry = clock_lo(); S.preset = -1; S.resumed = 0;
    }
    if (tid < (unsigned)(R * NDT_VW)) slots[tid / NDT_VW].cache[tid % NDT_VW].key = 0u;
    __syncthreads();                    // the only workgroup barrier of the kernel
#ifdef NDT_MATCH_TL
    if (tid == 0) atomicMin((unsigned long long *)&g_tl[4 * NDT_TL_PAIRS + 1024], (unsigned long long)wall_clock64());
#endif

    NdtMatchWork *work = reinterpret_cast<NdtMatchWork *>(work_mem);
    unsigned *ids = reinterpret_cast<unsigned *>(work_mem + sizeof(NdtMatchWork));
    const unsigned n_ids = n_pairs + gridDim.x * (unsigned)R + 1u;
    NdtParkedState *parked =
        reinterpret_cast<NdtParkedState *>(work_mem + ((sizeof(NdtMatchWork) + n_ids * sizeof(unsigned) + 7) & ~(size_t)7));
    double *const wsrc = w_src + wave * (9 * 64);
    uint2 *const wwin = w_win + wave * (7 * 64);

    unsigned idle_spins = 0u;
    for (;;) {
        // ---- find something to do: a share of a running evaluation, else a free slot to fill -------------------
        int task = TASK_NONE;
        bool running = false;           // a registration of this workgroup is between two evaluations: shares are about to appear
        if (lane == 0) {
#pragma unroll 1
            for (unsigned r = 0; r < (unsigned)R && task == TASK_NONE; r++) {
                const unsigned s = (wave + r) % (unsigned)R;
                Slot &S = slots[s];
                if (lds_load(&S.state) == SLOT_RUN) {
                    running = true;
                    if (lds_load(&S.next) < (unsigned)NDT_VW) {
                        const unsigned v = atomicAdd(&S.next, 1u);
                        if (v < (unsigned)NDT_VW) task = (int)(s * 16u + v);
                    }
                }
            }
            if (task == TASK_NONE) {
                const unsigned now = clock_lo();
#pragma unroll 1
                for (unsigned s = 0; s < (unsigned)R && task == TASK_NONE; s++) {
                    Slot &S = slots[s];
                    if (lds_load(&S.state) == SLOT_FREE && (int)(now - lds_load(&S.retry)) >= 0 &&
                        atomicCAS(&S.state, (unsigned)SLOT_FREE, (unsigned)SLOT_BUSY) == (unsigned)SLOT_FREE)
                        task = TASK_LOAD + (int)s;
                }
            }
            if (task == TASK_NONE && lds_load(&s_closed) >= (unsigned)R) task = TASK_EXIT;
        }
        task = __builtin_amdgcn_readfirstlane(task);
#ifdef NDT_MATCH_TL
        if (task == TASK_EXIT && lane == 0 && blockIdx.x < 1024u) atomicMax((unsigned long long *)&g_tl[4 * NDT_TL_PAIRS + blockIdx.x], (unsigned long long)wall_clock64());
#endif
        if (task == TASK_EXIT) return;
        if (task == TASK_NONE) {
            // (S_WAKEUP from the publishing wave would let idle waves sleep long and still react at once; on this
            //  hardware / runtime a kernel that executes it faults intermittently -- measured, 3 runs of 3 -- so the
            //  waves poll: at short intervals while a solver step of their workgroup is under way, else rarely)
            if (__builtin_amdgcn_readfirstlane(running ? 1 : 0)) __builtin_amdgcn_s_sleep(2);
            else __builtin_amdgcn_s_sleep(16);
            idle_spins += 1u;
            if ((idle_spins & 1023u) == 0u) {
                if (idle_spins > (1u << 24)) __hip_atomic_store(&work->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_load(&work->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
            }
            continue;
        }
        idle_spins = 0u;

        if (task >= TASK_LOAD) {
            // ---- fill a slot: the ticket drawn while parking, else a fresh pair, else a parked registration ----
            if (lane == 0) {
                const unsigned s = (unsigned)(task - TASK_LOAD);
                Slot &S = slots[s];
                unsigned new_state = SLOT_BUSY;
                while (new_state == SLOT_BUSY) {
                    int job = S.preset;         // >= 0: fresh pair, <= -2: resume parked pair (-2 - pair), -1: none
                    S.preset = -1;
                    if (job == -1) {
                        // slots beyond a workgroup's first draw once every workgroup has had its first ticket (a batch
                        // that cannot fill the chip keeps one registration per CU)
                        const unsigned f0 = __hip_atomic_load(&work->fresh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (f0 < n_pairs && (s == 0u || f0 >= gridDim.x)) {
                            const unsigned f = atomicAdd(&work->fresh, 1u);
                            if (f < n_pairs) job = (int)f;
                        }
                    }
                    while (job == -1) {
                        const unsigned f1 = __hip_atomic_load(&work->fresh, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned h = __hip_atomic_load(&work->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned rsv = __hip_atomic_load(&work->reserve, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (h >= rsv) {
                            // nothing parked; with the tickets gone nothing can be parked any more either
                            new_state = f1 >= n_pairs ? SLOT_CLOSED : SLOT_FREE;
                            break;
                        }
                        // a resumed registration is a long one: a workgroup runs one of them at a time unless more
                        // of them are waiting than other workgroups can take
                        if (lds_load(&s_resumed) != 0u && rsv - h <= double_thresh) { new_state = SLOT_FREE; break; }
                        if (atomicCAS(&work->head, h, h + 1u) != h) continue;
                        unsigned v, spins = 0u;
                        while ((v = __hip_atomic_load(&ids[h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
                            __builtin_amdgcn_s_sleep(8);
                            if (++spins > (1u << 22)) { __hip_atomic_store(&work->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v = 1u; break; }
                        }
                        if (v >= 2u) job = -2 - (int)(v - 2u);
                    }
                    if (job == -1) break;
                    const bool resumed = job <= -2;
                    const unsigned pair = resumed ? (unsigned)(-2 - job) : (unsigned)job;
                    // the indices and the maps come from device memory the host never saw: check them here
                    const uint32_t ti = tidx[pair], si = sidx[pair];
                    const bool bad_index = ti >= tset.n_maps || si >= sset.n_maps;
                    const bool truncated = !bad_index && (tset.counters[ti].overflow != 0u || sset.counters[si].overflow != 0u);
                    if (bad_index || truncated) {
                        NdtMatchResultDev *o = res + pair;      // (converged = 0; the pose is left untouched)
                        o->converged = 0; o->iterations = 0; o->fevals = 0;
                        o->exit_code = bad_index ? -2 : -3;     // -2 map index out of range, -3 a map needed more cells than max_cells
                        o->score = 0.0; o->n_source = 0; o->n_target = 0;
                        o->cycles_eval = 0; o->cycles_solver = 0; o->pair_terms_g = 0; o->pair_terms_h = 0;
                        continue;
                    }
                    S.tg = map_view(tset, ti);
                    S.sv = map_view(sset, si);
                    S.pair = pair;
                    if (resumed) {
                        const NdtParkedState &ps = parked[pair];
                        S.st = ps.st;
                        S.cnt[0] = ps.cnt[0]; S.cnt[1] = ps.cnt[1]; S.cnt[2] = ps.cnt[2]; S.cnt[3] = ps.cnt[3];
                    } else {
                        match_state_init(S.st, T16 + (size_t)pair * 16, s_prm, Q36 ? Q36 + (size_t)pair * 36 : nullptr);
                        S.cnt[0] = S.cnt[1] = S.cnt[2] = S.cnt[3] = 0ull;
                        if (S.st.done) { slot_result(S, T16, res); continue; }   // parameters the solver rejects
                    }
                    NDT_TL(pair, resumed ? 2 : 0)
                    S.n_feat = 0u; S.feat = nullptr;
                    if (feat_off) {
                        const unsigned f0 = feat_off[pair], f1 = feat_off[pair + 1u];
                        S.n_feat = f1 > f0 ? (f1 - f0 > 64u ? 64u : f1 - f0) : 0u;     // (the host refused more than 64)
                        S.feat = feat_cells + (size_t)f0 * 18;
                    }
                    S.st.use_feat = S.n_feat ? 1 : 0;
                    S.st.ls_joint = (S.n_feat && (s_prm.fusion_flags & 4) && !(s_prm.fusion_flags & 1)) ? 1 : 0;
                    S.resumed = resumed ? 1 : 0;
                    S.session = atomicAdd(&s_session, 1u) + 1u;      // (a workgroup never sees 2^32 registrations)
                    S.with_h = S.st.with_h;
                    S.done = 0u;
                    if (resumed) atomicAdd(&s_resumed, 1u);
                    new_state = SLOT_RUN;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (new_state == SLOT_RUN) {
#ifdef NDT_MATCH_TL
                    S.t_pub = __builtin_readcyclecounter(); if (!S.resumed) S.wall = 0ull;
#endif
                    lds_store(&S.state, SLOT_RUN);
                    lds_store(&S.next, 0u);
                } else {
                    if (new_state == SLOT_FREE) lds_store(&S.retry, clock_lo() + 2048u);
                    else atomicAdd(&s_closed, 1u);
                    lds_store(&S.state, new_state);
                }
            }
            continue;
        }

        // ---- one share of an evaluation ---------------------------------------------------------------------------
        Slot &S = slots[task >> 4];
        {
            const unsigned v = (unsigned)task & 15u;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");     // the request was published before `next`
            const long long c0 = __builtin_readcyclecounter();
            const int with_h = S.with_h;
            // (NDTMatcherD2D_2D without the Tikhonov term: the planar pair term, see pair_term)
            const bool planar = (s_prm.dof_mask & 0x3f) == 0x23 && !(s_prm.fusion_flags & 2);
            if (planar) {
                if (with_h) run_share<NN, true, QL, true>(S, v, wsrc, wwin, s_prm.lfd1, s_prm.lfd2);
                else run_share<NN, false, QL, true>(S, v, wsrc, wwin, s_prm.lfd1, s_prm.lfd2);
            } else if (with_h) run_share<NN, true, QL>(S, v, wsrc, wwin, s_prm.lfd1, s_prm.lfd2);
            else run_share<NN, false, QL>(S, v, wsrc, wwin, s_prm.lfd1, s_prm.lfd2);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // the row of partial sums before the count
            unsigned d = 0;
            if (lane == 0) {
                atomicAdd(&S.cnt[0], (unsigned long long)((long long)__builtin_readcyclecounter() - c0));
                d = atomicAdd(&S.done, 1u);
            }
            if ((unsigned)__builtin_amdgcn_readfirstlane((int)d) != (unsigned)NDT_VW - 1u) continue;
        }
        // ---- this wave delivered the last share: add the rows in share order and run the solver step (lane 0; nothing
        //      of the evaluation is live any more) ---------------------------------------------------------------
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane < 29u) {
            double a = 0;
#pragma unroll
            for (int k = 0; k < NDT_VW; k++) a += S.part[k * 32 + lane];
            S.sums[lane] = a;             // [28]: number of (source, target) pair terms of this evaluation
        }
        ndt_wave_sync();
        const long long c1 = __builtin_readcyclecounter();
        slot_step(S, s_prm);
        ndt_wave_sync();
        if (lane == 0) {
#ifdef NDT_MATCH_TL
            S.wall += (unsigned long long)(c1 - S.t_pub);
#endif
            S.cnt[S.with_h ? 3 : 2] += (unsigned long long)S.sums[28];
            S.cnt[1] += (unsigned long long)((long long)__builtin_readcyclecounter() - c1);
            bool release = false;
            if (S.st.done) {
                slot_result(S, T16, res);
                NDT_TL(S.pair, 3)
                release = true;
            } else if (S.st.phase == PH_NEWTON && park_iters > 0 && S.st.itr_ctr >= park_iters &&
                       __hip_atomic_load(&work->fresh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_pairs) {
                // about to start another Newton iteration of a long registration: hand the slot to a pair that has
                // not started yet, if there is one
                const unsigned pair = S.pair;
                const unsigned slot = atomicAdd(&work->reserve, 1u);
                const unsigned f = atomicAdd(&work->fresh, 1u);
                if (f < n_pairs) {
                    NdtParkedState &ps = parked[pair];
                    ps.st = S.st;
                    ps.cnt[0] = S.cnt[0]; ps.cnt[1] = S.cnt[1]; ps.cnt[2] = S.cnt[2]; ps.cnt[3] = S.cnt[3];
                    __hip_atomic_store(&ids[slot], pair + 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    NDT_TL(pair, 1)
                    S.preset = (int)f;
                    release = true;
                } else {
                    __hip_atomic_store(&ids[slot], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            lds_store(&S.done, 0u);
            if (release) {
                if (S.resumed) atomicSub(&s_resumed, 1u);
                lds_store(&S.retry, clock_lo());
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                lds_store(&S.state, SLOT_FREE);
            } else {
                S.with_h = S.st.with_h;      // the next request: pose in S.st.Teval, with / without Hessian
#ifdef NDT_MATCH_TL
                S.t_pub = __builtin_readcyclecounter();
#endif
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                lds_store(&S.next, 0u);
            }
        }
    }
}

// ---- the stream-fed matcher (round 5) ----------------------------------------------------------------------------------
// The registrar's pipelined form of the kernel above.  A launch of ndt_match_kernel ends when its longest registration does, and
// every batch of 1024 pairs has ~50 of them (ITR_MAX): two thirds of a launch is a tail on a fifth of its CUs, which the
// registrar could only hide behind the neighbouring batches' launches (three streams, 2.1 ms per batch where the chip's
// work is 1.8).  Here ONE running instance serves batch after batch: batches are published into a queue in device memory (a
// one-thread kernel behind the batch's grid builds, on the build stream), a slot that finds its batch out of tickets moves on
// to the next published one, and long registrations simply keep their slots while their neighbours work through later
// batches -- no parking, no tail per batch, one tail per run.  The instance holds G < all CUs (one workgroup each, as
// above); the grid builds of the next batches run on the rest of the chip, concurrently, for as long as there is work.
//   * An instance exits when every published batch is complete and nothing is left to draw; every publish is followed by an
//     instance launch on the matcher stream (it starts when the running one has ended, or finds its batch taken and leaves).
//   * Memory: the maps of a batch are written by build kernels that have ENDED (stream order) before the batch is published;
//     a slot that moves to a new batch does a system-scope acquire (L1 / L2 invalidate: the map set of a ring entry is reused
//     every `depth` batches, stale lines may be cached).  Poses and results are written back with a system-scope release before
//     a registration is counted as done; queue words are system-scope atomics.
//   * A registration's arithmetic is that of ndt_match_kernel -- same shares, same order of sums: the same bits.
//   * Two registrations per workgroup, as above.  Measured with three (half the hit list each): best 562 k registrations/s at 128
//     workgroups against 578 k with two at 144 (100 steps of the bench).
#define NDT_STREAM_RING 8
#define NDT_STREAM_STAMPS 64
struct NdtStreamBatch {
    NdtSetView set;                  // targets in maps [0, n_pairs), sources in [n_pairs, 2 n_pairs)
    double *T16;
    NdtMatchResultDev *res;
    NdtMatchParamsDev prm;
    unsigned n_pairs, seq;
    // (seq + 1) << 32 | tickets drawn.  The generation names the batch a ticket belongs to: a draw that lands on an entry that
    // is being re-published (a slot held up for longer than a whole batch takes) carries the OLD generation, is recognised
    // by that and thrown away -- the publisher's single store of the new word erases it without losing a ticket of the new
    // batch (zeroing `fresh` in two steps could hand a ticket out twice).
    unsigned long long fresh;
    unsigned done, pad_;             // registrations finished
};
struct NdtStreamQueue {
    unsigned published;              // batches published so far (their descriptors are complete)
    unsigned first_open;             // batches below this one have no tickets left
    unsigned completed;              // batches whose registrations have all finished
    unsigned abort;
    unsigned ring;                   // entries in use (= the registrar's depth: entry e always describes map set e; set by the host)
    unsigned linger;                 // 100 MHz ticks an instance that has worked stays when it runs dry (set by the host)
    unsigned live;                   // workgroups of matcher instances that are resident
    unsigned final_pub;              // the host is waiting for everything: an instance that finds `published` at this value complete
                                     // does not linger (ndtgpu_registrar_sync)
    unsigned done_seq[NDT_STREAM_RING];   // ring entry e: seq + 1 of the last batch that completed in it
    // 100 MHz time stamps of the last NDT_STREAM_STAMPS batches: [seq % N][0] published, [1] last registration finished
    // (ndtgpu_registrar_kernel_ms: the matcher side of a sub-batch as the queue saw it)
    unsigned long long stamp[NDT_STREAM_STAMPS][2];
    NdtStreamBatch b[NDT_STREAM_RING];
};
size_t ndt_stream_queue_bytes() { return sizeof(NdtStreamQueue); }
size_t ndt_stream_abort_offset() { return offsetof(NdtStreamQueue, abort); }
size_t ndt_stream_ring_offset() { return offsetof(NdtStreamQueue, ring); }     // {ring, linger}: two words the host sets
unsigned ndt_stream_ring() { return NDT_STREAM_RING; }
unsigned ndt_stream_stamps() { return NDT_STREAM_STAMPS; }
// After an abort (and with every stream of the registrar idle): the queue as if all `submitted` batches were complete.
hipError_t ndt_stream_reset(void *queue_dev, unsigned submitted, unsigned ring)
{
    struct { unsigned published, first_open, completed, abort; } head = {submitted, submitted, submitted, 0u};
    hipError_t e = hipMemcpy(queue_dev, &head, sizeof head, hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    unsigned live = 0u, done_seq[NDT_STREAM_RING] = {};
    for (unsigned j = submitted > ring ? submitted - ring : 0u; j < submitted; j++) done_seq[j % ring] = j + 1u;
    e = hipMemcpy((char *)queue_dev + offsetof(NdtStreamQueue, live), &live, sizeof live, hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    return hipMemcpy((char *)queue_dev + offsetof(NdtStreamQueue, done_seq), done_seq, sizeof done_seq, hipMemcpyHostToDevice);
}
// the two stamps of batch `seq` (valid while fewer than NDT_STREAM_STAMPS batches have been published since)
size_t ndt_stream_live_offset() { return offsetof(NdtStreamQueue, live); }
hipError_t ndt_stream_read_stamps(const void *queue_dev, unsigned seq, unsigned long long out[2])
{
    return hipMemcpy(out, (const char *)queue_dev + offsetof(NdtStreamQueue, stamp) + (size_t)(seq % NDT_STREAM_STAMPS) * 16u, 16u,
                     hipMemcpyDeviceToHost);
}

namespace {
NDT_D unsigned sys_load(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
NDT_D void sys_store(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
struct StreamSlotExt {               // what a slot of the stream-fed matcher knows about the batch its registration belongs to
    NdtMatchParamsDev prm;
    double *T16;
    NdtMatchResultDev *res;
    unsigned *done_ctr;
    unsigned n_pairs, seq;
    unsigned cur;                    // the batch this slot draws its next ticket from
    unsigned worked;                 // this slot has run a registration
    unsigned dry_since;              // 100 MHz clock (low word) when the slot first found everything complete (0: not dry)
};
}  // namespace

// 256 threads.  Every result of the batch starts as "not run" (exit_code -4, converged 0): a batch that an abort cuts short
// describes itself, registration by registration.  Then the descriptor -- everything but the ticket word --, and the ticket
// word (new generation, no tickets drawn) with ONE store: from then on draws are valid.
__global__ void ndt_stream_publish_kernel(NdtStreamQueue *q, NdtStreamBatch desc)
{
    for (unsigned i = threadIdx.x; i < desc.n_pairs; i += blockDim.x) {
        NdtMatchResultDev *o = desc.res + i;
        o->converged = 0; o->iterations = 0; o->fevals = 0; o->exit_code = -4;
        o->score = 0.0; o->n_source = 0; o->n_target = 0;
        o->cycles_eval = 0; o->cycles_solver = 0; o->pair_terms_g = 0; o->pair_terms_h = 0;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x != 0) return;
    NdtStreamBatch *b = &q->b[desc.seq % q->ring];
    b->set = desc.set; b->T16 = desc.T16; b->res = desc.res; b->prm = desc.prm;
    b->n_pairs = desc.n_pairs; b->seq = desc.seq;
    b->done = 0u;
    q->stamp[desc.seq % NDT_STREAM_STAMPS][0] = wall_clock64();
    q->stamp[desc.seq % NDT_STREAM_STAMPS][1] = 0ull;
    __threadfence_system();
    __hip_atomic_store(&b->fresh, (unsigned long long)(desc.seq + 1u) << 32, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&q->published, desc.seq + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// blocks its stream until ring entry `entry` has completed batch seq >= value - 1.  Bounded by PROGRESS, not by time: it gives
// up (and raises the abort word) only when no registration of the queue has finished for ~30 s (a cold box pages code objects
// in for seconds: the first profiled run of round 5 lost a batch to a 2 s bound) -- a caller may wait for a
// batch that hundreds of others are queued in front of.
__global__ void ndt_stream_wait_kernel(NdtStreamQueue *q, unsigned entry, unsigned value)
{
    unsigned spins = 0u, seen = 0u;
    bool gave_up = false;
    while ((int)(sys_load(&q->done_seq[entry]) - value) < 0) {
        __builtin_amdgcn_s_sleep(32);
        if ((++spins & 255u) == 0u) {
            unsigned progress = sys_load(&q->completed);
            for (unsigned e = 0; e < q->ring; e++) progress += sys_load(&q->b[e].done);
            if (progress != seen) { seen = progress; spins = 0u; }
            if (spins > (1u << 25)) sys_store(&q->abort, 2u);                 // ~30 s without any progress
            if (sys_load(&q->abort)) { gave_up = true; break; }
        }
    }
    // What follows this kernel on its stream rebuilds a map set (or reads a batch's outputs).  After an abort the batch is not
    // complete: the instances stop drawing tickets, but registrations that are under way still read their maps -- the stream is
    // held until every resident workgroup has left (they finish what they hold: milliseconds; bounded all the same).
    if (gave_up)
        for (unsigned k = 0; k < (1u << 22) && sys_load(&q->live) != 0u; k++) __builtin_amdgcn_s_sleep(64);
}

// R registrations in flight per workgroup: two with hit lists of 1024 entries per share, or THREE with 640 (what 160 KB of LDS
// hold) -- enough for maps of up to ~450 cells (62 cells x 5 neighbours per share), whose registrations spend 40 % of their time
// in one-lane solver steps that a third registration covers (round 6, bench: 666 -> 722-738 k registrations/s; the cluttered scene,
// 1 085 hits per share in two passes: 88-92 against 99 k -- the registrar picks by the cells per map it measured; four with lists
// of 384: 660 k).  Same shares, same sums: the same bits.
template <int NN, int R>
__global__ __launch_bounds__(NDT_MATCH_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void ndt_match_stream_kernel(NdtStreamQueue *q)
{
#ifndef NDT_STREAM_RETRY
#define NDT_STREAM_RETRY 4096
#endif
#ifndef NDT_STREAM_QLX
#define NDT_STREAM_QLX 640
#endif
#ifndef NDT_STREAM_RX
#define NDT_STREAM_RX 3
#endif
    constexpr int QL = R >= 3 ? NDT_STREAM_QLX : 1024;
    typedef MatchSlot<QL> Slot;
    __shared__ Slot slots[R];
    __shared__ StreamSlotExt ext[R];
    __shared__ double w_src[NDT_MATCH_WAVES * 9 * 64];
    __shared__ uint2 w_win[NDT_MATCH_WAVES * 7 * 64];
    __shared__ unsigned s_session, s_closed, s_seen, s_left;

    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const unsigned ring = sys_load(&q->ring);
    if (tid == 0) {
        s_session = 0u; s_closed = 0u; s_seen = 0u; s_left = 0u;
        __hip_atomic_fetch_add(&q->live, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (tid < (unsigned)R) {
        Slot &S = slots[tid];
        S.state = SLOT_FREE; S.next = NDT_VW; S.done = 0u; S.retry = clock_lo(); S.preset = -1; S.resumed = 0;
        S.n_feat = 0u; S.feat = nullptr;
        ext[tid].cur = 0u; ext[tid].worked = 0u; ext[tid].dry_since = 0u;
    }
    const unsigned linger = sys_load(&q->linger);
    if (tid < (unsigned)(R * NDT_VW)) slots[tid / NDT_VW].cache[tid % NDT_VW].key = 0u;
    __syncthreads();
    double *const wsrc = w_src + wave * (9 * 64);
    uint2 *const wwin = w_win + wave * (7 * 64);

    unsigned idle_spins = 0u;
    for (;;) {
        int task = TASK_NONE;
        bool running = false;
        if (lane == 0) {
#pragma unroll 1
            for (unsigned r = 0; r < (unsigned)R && task == TASK_NONE; r++) {
                const unsigned s = (wave + r) % (unsigned)R;
                Slot &S = slots[s];
                if (lds_load(&S.state) == SLOT_RUN) {
                    running = true;
                    if (lds_load(&S.next) < (unsigned)NDT_VW) {
                        const unsigned v = atomicAdd(&S.next, 1u);
                        if (v < (unsigned)NDT_VW) task = (int)(s * 16u + v);
                    }
                }
            }
            if (task == TASK_NONE) {
                const unsigned now = clock_lo();
#pragma unroll 1
                for (unsigned s = 0; s < (unsigned)R && task == TASK_NONE; s++) {
                    Slot &S = slots[s];
                    if (lds_load(&S.state) == SLOT_FREE && (int)(now - lds_load(&S.retry)) >= 0 &&
                        atomicCAS(&S.state, (unsigned)SLOT_FREE, (unsigned)SLOT_BUSY) == (unsigned)SLOT_FREE)
                        task = TASK_LOAD + (int)s;
                }
            }
            if (task == TASK_NONE && lds_load(&s_closed) >= (unsigned)R) task = TASK_EXIT;
        }
        task = __builtin_amdgcn_readfirstlane(task);
        if (task == TASK_EXIT) {
            // (the last wave out takes the workgroup off the queue's count of resident workgroups)
            if (lane == 0 && atomicAdd(&s_left, 1u) == (unsigned)NDT_MATCH_WAVES - 1u)
                __hip_atomic_fetch_sub(&q->live, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        if (task == TASK_NONE) {
            if (__builtin_amdgcn_readfirstlane(running ? 1 : 0)) __builtin_amdgcn_s_sleep(2);
            else __builtin_amdgcn_s_sleep(16);
            idle_spins += 1u;
            if ((idle_spins & 1023u) == 0u) {
                if (idle_spins > (1u << 26)) sys_store(&q->abort, 1u);          // ~30 s without work
                // (an abort: nothing new is drawn -- the loader below closes the slots.  A wave leaves when no registration of its
                //  workgroup is under way; registrations that are finish first, a second at most)
                if (sys_load(&q->abort) && (!__builtin_amdgcn_readfirstlane(running ? 1 : 0) || idle_spins > (1u << 22))) {
                    if (lane == 0 && atomicAdd(&s_left, 1u) == (unsigned)NDT_MATCH_WAVES - 1u)
                        __hip_atomic_fetch_sub(&q->live, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    return;
                }
            }
            continue;
        }
        idle_spins = 0u;

        if (task >= TASK_LOAD) {
            // ---- fill a slot: the next ticket of the slot's batch, else of the next published batch ------------------------
            if (lane == 0) {
                const unsigned s = (unsigned)(task - TASK_LOAD);
                Slot &S = slots[s];
                StreamSlotExt &E = ext[s];
                unsigned new_state = SLOT_BUSY;
                while (new_state == SLOT_BUSY) {
                    unsigned c = E.cur;
                    // (the queue words in ONE round trip: system-scope loads are issued in program order and waited for where
                    //  they are first used.  `published` is a plain load here: the acquire that goes with it is the fence below,
                    //  once per workgroup and batch -- as an acquire LOAD it invalidated this CU's L1 for every registration)
                    const unsigned fo = sys_load(&q->first_open);
                    const unsigned pub = sys_load(&q->published);
                    const unsigned comp = sys_load(&q->completed);
                    if (sys_load(&q->abort)) { new_state = SLOT_CLOSED; break; }     // nothing new after an abort
                    if ((int)(c - fo) < 0) c = fo;
                    E.cur = c;
                    if ((int)(c - pub) >= 0) {
                        // nothing to draw.  Everything published complete as well: this instance is done (a batch published
                        // later brings its own launch); else the tails of others are running: look again soon
                        // (an instance that has worked stays for `linger`: when the builds are the slower side a batch is
                        //  complete before the next one is published, and CUs given up now are taken by that build's workgroups,
                        //  behind which the next instance would have to queue)
                        new_state = SLOT_FREE;
                        if (comp == pub) {
                            const unsigned now = (unsigned)wall_clock64() | 1u;
                            if (!E.worked || linger == 0u || sys_load(&q->final_pub) == pub) new_state = SLOT_CLOSED;
                            else if (E.dry_since == 0u) E.dry_since = now;
                            else if (now - E.dry_since > linger) new_state = SLOT_CLOSED;
                        } else {
                            E.dry_since = 0u;
                        }
                        break;
                    }
                    NdtStreamBatch *B = &q->b[c % ring];
                    // (what changes from batch to batch in a ring entry is read past the caches; B->set never changes)
                    // The ticket first, the descriptor after it: a ring entry is only re-published once its batch is complete,
                    // so whoever holds a ticket below n_pairs reads the descriptor of the batch the ticket belongs to -- and
                    // takes the batch's number from there (a workgroup that was held up between the queue words and the draw
                    // for longer than a whole batch takes would otherwise file the registration under the old number).
                    const volatile NdtStreamBatch *Bv = B;
                    const unsigned long long ticket = __hip_atomic_fetch_add(&B->fresh, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    const unsigned f = (unsigned)ticket;
                    // (the whole descriptor in one round trip, before the first of its words is looked at)
                    const unsigned n = Bv->n_pairs, bseq = Bv->seq;
                    const int d_nn = Bv->prm.n_neighbours, d_itr = Bv->prm.itr_max, d_sc = Bv->prm.step_control, d_dof = Bv->prm.dof_mask;
                    const int d_uig = Bv->prm.use_initial_guess, d_ff = Bv->prm.fusion_flags;
                    const double d_ds = Bv->prm.delta_score, d_l1 = Bv->prm.lfd1, d_l2 = Bv->prm.lfd2;
                    double *const d_T16 = Bv->T16;
                    NdtMatchResultDev *const d_res = Bv->res;
                    // a ticket of another generation than the descriptor's: the entry is being re-published (this slot was held up
                    // for longer than a whole batch takes).  The draw is void -- the publisher's store erases it -- : look again
                    if ((unsigned)(ticket >> 32) != bseq + 1u) continue;
                    if (f >= n) {
                        if (bseq == c) {
                            __hip_atomic_fetch_max(&q->first_open, c + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            E.cur = c + 1u;
                        }
                        continue;
                    }
                    c = bseq; E.cur = c;
                    // ONE acquire per workgroup and batch (batches are taken in order: whoever of the workgroup's slots reaches
                    // batch c first does it -- the invalidate also drops the maps its XCD's other registrations are reading)
                    if ((int)(lds_load(&s_seen) - (c + 1u)) < 0) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // (system scope: this CU's L1, this XCD's L2)
                        lds_store(&s_seen, c + 1u);
                    }
                    E.prm.n_neighbours = d_nn; E.prm.itr_max = d_itr; E.prm.step_control = d_sc;
                    E.prm.dof_mask = d_dof; E.prm.use_initial_guess = d_uig;
                    E.prm.fusion_flags = d_ff; E.prm.delta_score = d_ds;
                    E.prm.lfd1 = d_l1; E.prm.lfd2 = d_l2;
                    E.T16 = d_T16; E.res = d_res; E.done_ctr = &B->done; E.n_pairs = n; E.seq = c;
                    const unsigned pair = f, ti = f, si = n + f;
                    bool finished = false;
                    if (B->set.counters[ti].overflow != 0u || B->set.counters[si].overflow != 0u) {
                        NdtMatchResultDev *o = E.res + pair;            // (converged = 0; the pose is left untouched)
                        o->converged = 0; o->iterations = 0; o->fevals = 0; o->exit_code = -3;
                        o->score = 0.0; o->n_source = 0; o->n_target = 0;
                        o->cycles_eval = 0; o->cycles_solver = 0; o->pair_terms_g = 0; o->pair_terms_h = 0;
                        finished = true;
                    } else {
                        S.tg = map_view(B->set, ti);
                        S.sv = map_view(B->set, si);
                        S.pair = pair;
                        match_state_init(S.st, E.T16 + (size_t)pair * 16, E.prm, nullptr);
                        S.cnt[0] = S.cnt[1] = S.cnt[2] = S.cnt[3] = 0ull;
                        if (S.st.done) { slot_result(S, E.T16, E.res); finished = true; }   // parameters the solver rejects
                    }
                    if (finished) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                        const unsigned d = __hip_atomic_fetch_add(E.done_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (d + 1u == n) {
                            __hip_atomic_fetch_max(&q->first_open, c + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            q->stamp[c % NDT_STREAM_STAMPS][1] = wall_clock64();
                            sys_store(&q->done_seq[c % ring], c + 1u);
                            __hip_atomic_fetch_add(&q->completed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                        continue;
                    }
                    S.st.use_feat = 0; S.st.ls_joint = 0;
                    E.worked = 1u; E.dry_since = 0u;
                    S.session = atomicAdd(&s_session, 1u) + 1u;
                    S.with_h = S.st.with_h;
                    S.done = 0u;
                    new_state = SLOT_RUN;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (new_state == SLOT_RUN) {
                    lds_store(&S.state, SLOT_RUN);
                    lds_store(&S.next, 0u);
                } else {
                    if (new_state == SLOT_FREE) lds_store(&S.retry, clock_lo() + (unsigned)NDT_STREAM_RETRY);
                    else atomicAdd(&s_closed, 1u);
                    lds_store(&S.state, new_state);
                }
            }
            continue;
        }

        // ---- one share of an evaluation ---------------------------------------------------------------------------------
        Slot &S = slots[task >> 4];
        StreamSlotExt &E = ext[task >> 4];
        {
            const unsigned v = (unsigned)task & 15u;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const long long c0 = __builtin_readcyclecounter();
            const int with_h = S.with_h;
            const bool planar = (E.prm.dof_mask & 0x3f) == 0x23 && !(E.prm.fusion_flags & 2);
            if (planar) {
                if (with_h) run_share<NN, true, QL, true>(S, v, wsrc, wwin, E.prm.lfd1, E.prm.lfd2);
                else run_share<NN, false, QL, true>(S, v, wsrc, wwin, E.prm.lfd1, E.prm.lfd2);
            } else if (with_h) run_share<NN, true, QL>(S, v, wsrc, wwin, E.prm.lfd1, E.prm.lfd2);
            else run_share<NN, false, QL>(S, v, wsrc, wwin, E.prm.lfd1, E.prm.lfd2);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            unsigned d = 0;
            if (lane == 0) {
                atomicAdd(&S.cnt[0], (unsigned long long)((long long)__builtin_readcyclecounter() - c0));
                d = atomicAdd(&S.done, 1u);
            }
            if ((unsigned)__builtin_amdgcn_readfirstlane((int)d) != (unsigned)NDT_VW - 1u) continue;
        }
        // ---- last share delivered: rows in share order, solver step ---------------------------------------------------------
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane < 29u) {
            double a = 0;
#pragma unroll
            for (int k = 0; k < NDT_VW; k++) a += S.part[k * 32 + lane];
            S.sums[lane] = a;
        }
        ndt_wave_sync();
        const long long c1 = __builtin_readcyclecounter();
        slot_step(S, E.prm);
        ndt_wave_sync();
        if (lane == 0) {
            S.cnt[S.with_h ? 3 : 2] += (unsigned long long)S.sums[28];
            S.cnt[1] += (unsigned long long)((long long)__builtin_readcyclecounter() - c1);
            lds_store(&S.done, 0u);
            if (S.st.done) {
                slot_result(S, E.T16, E.res);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");              // (system scope: pose and result reach memory)
                const unsigned d = __hip_atomic_fetch_add(E.done_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (d + 1u == E.n_pairs) {
                    // (a complete batch has no tickets left: nobody looks at its ring entry again, it may be re-published)
                    __hip_atomic_fetch_max(&q->first_open, E.seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    q->stamp[E.seq % NDT_STREAM_STAMPS][1] = wall_clock64();
                    sys_store(&q->done_seq[E.seq % ring], E.seq + 1u);
                    __hip_atomic_fetch_add(&q->completed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                lds_store(&S.retry, clock_lo());
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                lds_store(&S.state, SLOT_FREE);
            } else {
                S.with_h = S.st.with_h;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                lds_store(&S.next, 0u);
            }
        }
    }
}

// a batch that was registered outside the queue (the registrar's calibration batch): the queue counts it as published and done
__global__ void ndt_stream_skip_kernel(NdtStreamQueue *q, unsigned seq)
{
    sys_store(&q->first_open, seq + 1u);
    sys_store(&q->completed, seq + 1u);
    sys_store(&q->done_seq[seq % q->ring], seq + 1u);
    __threadfence_system();
    __hip_atomic_store(&q->published, seq + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// "nothing more is coming until the host says so": instances that find batch `published` - 1 complete leave without lingering
__global__ void ndt_stream_final_kernel(NdtStreamQueue *q, unsigned published) { sys_store(&q->final_pub, published); }
hipError_t ndt_stream_final(void *queue_dev, unsigned published, hipStream_t stream)
{
    hipLaunchKernelGGL(ndt_stream_final_kernel, dim3(1), dim3(1), 0, stream, (NdtStreamQueue *)queue_dev, published);
    return hipGetLastError();
}
hipError_t ndt_stream_skip(void *queue_dev, unsigned seq, hipStream_t stream)
{
    hipLaunchKernelGGL(ndt_stream_skip_kernel, dim3(1), dim3(1), 0, stream, (NdtStreamQueue *)queue_dev, seq);
    return hipGetLastError();
}

hipError_t ndt_stream_publish(void *queue_dev, const NdtSetView &set, double *T16_dev, NdtMatchResultDev *res_dev,
                              const NdtMatchParamsDev &prm, unsigned n_pairs, unsigned seq, hipStream_t stream)
{
    NdtStreamBatch d;
    d.set = set; d.T16 = T16_dev; d.res = res_dev; d.prm = prm; d.n_pairs = n_pairs; d.seq = seq; d.fresh = 0ull; d.done = 0u; d.pad_ = 0u;
    hipLaunchKernelGGL(ndt_stream_publish_kernel, dim3(1), dim3(256), 0, stream, (NdtStreamQueue *)queue_dev, d);
    return hipGetLastError();
}

hipError_t ndt_stream_wait(void *queue_dev, unsigned ring, unsigned seq, hipStream_t stream)
{
    hipLaunchKernelGGL(ndt_stream_wait_kernel, dim3(1), dim3(1), 0, stream, (NdtStreamQueue *)queue_dev, seq % ring, seq + 1u);
    return hipGetLastError();
}

hipError_t ndt_launch_match_stream(void *queue_dev, int n_neighbours, int slots, unsigned n_groups, hipStream_t stream)
{
    NdtStreamQueue *q = (NdtStreamQueue *)queue_dev;
#define NDT_LAUNCH_STREAM(NN_)                                                                                                   \
    do {                                                                                                                         \
        if (slots >= 3) hipLaunchKernelGGL((ndt_match_stream_kernel<NN_, NDT_STREAM_RX>), dim3(n_groups), dim3(NDT_MATCH_THREADS), 0, stream, q); \
        else hipLaunchKernelGGL((ndt_match_stream_kernel<NN_, 2>), dim3(n_groups), dim3(NDT_MATCH_THREADS), 0, stream, q);         \
    } while (0)
    switch (n_neighbours) {
    case 0: NDT_LAUNCH_STREAM(0); break;
    case 1: NDT_LAUNCH_STREAM(1); break;
    case 2: NDT_LAUNCH_STREAM(2); break;
    case 3: NDT_LAUNCH_STREAM(3); break;
    default: return hipErrorInvalidValue;
    }
#undef NDT_LAUNCH_STREAM
    return hipGetLastError();
}

// NDTMatcherD2D::derivativesNDT as a stand-alone entry (host-driven matchFusion loop, FD tests).
template <int NN>
__global__ __launch_bounds__(NDT_MATCH_THREADS) void ndt_derivatives_kernel(
    NdtSetView tset, unsigned tmap, const NdtCell *__restrict__ src, unsigned m, int with_h, double lfd1,
    double lfd2, double *__restrict__ out28)
{
    __shared__ EvalShared<NDT_MATCH_WAVES> sh;
    const MapView tg = map_view(tset, tmap);
    rigid I;
    for (int k = 0; k < 9; k++) I.r[k] = (k % 4 == 0) ? 1.0 : 0.0;
    I.t[0] = I.t[1] = I.t[2] = 0.0;
    if (threadIdx.x < 28) sh.sums[threadIdx.x] = 0.0;
    __syncthreads();
    if (with_h) eval_derivs<NN, true>(tg, (gcell_ptr)src, (int)m, I, lfd1, lfd2, sh);
    else eval_derivs<NN, false>(tg, (gcell_ptr)src, (int)m, I, lfd1, lfd2, sh);
    if (threadIdx.x < 28) out28[threadIdx.x] = sh.sums[threadIdx.x];
}

// NDTMatcherD2D::covariance(target, source, T, cov) (ndt_feature_graph.cpp:296-298; fuser_hmt.cpp:403-405) for n
// links, one workgroup per link: cov = H^-1 (sigma_S J^T J) H^-1 with H the D2D Hessian at T (the evaluation above)
// and one row of J per source cell whose transformed mean falls into a target cell with a Gaussian (perception_oru,
// SURVEY.md App. A.7; mode 0: the cell's own pose Jacobians, mode 1: the matcher's constructor values j = [I 0], Z = 0).
template <int NN>
__global__ __launch_bounds__(NDT_MATCH_THREADS) void ndt_covariance_kernel(
    NdtSetView tset, const uint32_t *__restrict__ tidx, NdtSetView sset, const uint32_t *__restrict__ sidx,
    const double *__restrict__ T16, double lfd1, double lfd2, int mode, double *__restrict__ cov36,
    int *__restrict__ status)
{
    __shared__ EvalShared<NDT_MATCH_WAVES> sh;
    __shared__ rigid s_T;
    const unsigned link = blockIdx.x, tid = threadIdx.x, wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), lane = tid & 63u;
    const MapView tg = map_view(tset, tidx[link]);
    const MapView sv = map_view(sset, sidx[link]);
    const uint2 *trank = tset.rankmap + (size_t)tidx[link] * ndt_rm_stride(tset.grid);
    if (tid == 0) {
        const double *Tm = T16 + (size_t)link * 16;
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) s_T.r[r * 3 + c] = Tm[c * 4 + r];
            s_T.t[r] = Tm[12 + r];
        }
    }
    __syncthreads();
    eval_derivs<NN, true>(tg, sv.cells, sv.n_cells, s_T, lfd1, lfd2, sh);
    // (the Hessian stays in sh.sums[7..27]: the pass below only writes sh.part)
    // J^T J: 21 sums over the source cells
    const double sigmaS = 0.03 * 0.03;
    double jj[32];
#pragma unroll
    for (int k = 0; k < 32; k++) jj[k] = 0.0;
    const rigid T = s_T;
    for (int i = (int)tid; i < sv.n_cells; i += NDT_MATCH_THREADS) {
        gcell_ptr sc = sv.cells + i;
        const d3 m = apply(T, d3{sc->mean[0], sc->mean[1], sc->mean[2]});
        const sym3 C = rotate_cov(T.r, sym3{sc->cov[0], sc->cov[1], sc->cov[2], sc->cov[3], sc->cov[4], sc->cov[5]});
        const int ix = lazygrid_index(m.x, tg.cx, tg.res, tg.sx), iy = lazygrid_index(m.y, tg.cy, tg.res, tg.sy),
                  iz = lazygrid_index(m.z, tg.cz, tg.res, tg.sz);
        if ((unsigned)ix >= (unsigned)tg.sx || (unsigned)iy >= (unsigned)tg.sy || (unsigned)iz >= (unsigned)tg.sz) continue;
        const int r = ndt_rank_of(trank, (unsigned)((ix * tg.sy + iy) * tg.sz + iz));
        if (r < 0) continue;
        gcell_ptr tc = tg.cells + r;
        const d3 x = m - d3{tc->mean[0], tc->mean[1], tc->mean[2]};
        sym3 B;
        if (!inverse_check(C + sym3{tc->cov[0], tc->cov[1], tc->cov[2], tc->cov[3], tc->cov[4], tc->cov[5]}, B)) continue;
        const d3 Bx = mul(B, x);
        double factor = -dot(x, Bx) / 2;
        if (factor < -120) continue;
        factor = exp(lfd2 * factor) / 2;
        if (factor > 1 || factor < 0 || factor * 0 != 0) continue;
        // Q = -sigma_S B B (symmetric): Q x = -sigma_S B (B x)
        const d3 Qx = (-sigmaS) * mul(B, Bx);
        const double f1 = dot(x, Qx);
        double G[6] = {Qx.x, Qx.y, Qx.z, 0.0, 0.0, 0.0};     // x^T Q j_a, translations
        if (mode == 0) {
            // rotations: j_a = e_a x m;  Z_a = [e_a]x C + C [e_a]x^T  =>  Z_a v = e_a x (C v) - C (e_a x v)
            const d3 CBx = mul(C, Bx), CQx = mul(C, Qx);
            const d3 ja[3] = {ex_cross(m), ey_cross(m), ez_cross(m)};
            const d3 zb[3] = {ex_cross(CBx) - mul(C, ex_cross(Bx)), ey_cross(CBx) - mul(C, ey_cross(Bx)), ez_cross(CBx) - mul(C, ez_cross(Bx))};
            const d3 zq[3] = {ex_cross(CQx) - mul(C, ex_cross(Qx)), ey_cross(CQx) - mul(C, ey_cross(Qx)), ez_cross(CQx) - mul(C, ez_cross(Qx))};
#pragma unroll
            for (int a = 0; a < 3; a++) G[3 + a] = dot(Qx, ja[a]) - dot(Qx, zb[a]) - dot(Bx, zq[a]);
        }
        const double sc2 = factor * lfd1 * lfd2 / 2;
#pragma unroll
        for (int a = 0; a < 6; a++) G[a] = (G[a] + (-lfd2 / 2) * f1) * sc2;
        int o = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = a; b < 6; b++) jj[o++] += G[a] * G[b];
    }
    {
        const double tot = wave_sum_all<32>(jj);
        if ((lane & 1u) == 0u && (lane >> 1) < 21u) sh.part[wave * 32 + (lane >> 1)] = tot;
    }
    __syncthreads();
    // H^-1 (sigma_S J^T J) H^-1 on thread 0; the matrices live in LDS (the pivot search indexes them dynamically:
    // as local arrays they would go to scratch memory)
    __shared__ double s_H[6][12], s_JK[6][6], s_tmp[6][6];
    if (tid == 0) {
        int o = 0;
        for (int a = 0; a < 6; a++)
            for (int b = a; b < 6; b++) {
                double s2 = 0;
                for (int k = 0; k < NDT_MATCH_WAVES; k++) s2 += sh.part[k * 32 + o];
                s_JK[a][b] = s_JK[b][a] = sigmaS * s2;
                s_H[a][b] = s_H[b][a] = sh.sums[7 + o];
                o++;
            }
        // H^-1 by Gauss-Jordan with partial pivoting (Eigen: cov.inverse())
        bool ok = true;
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < 6; b++) s_H[a][6 + b] = (a == b) ? 1.0 : 0.0;
        for (int c = 0; c < 6 && ok; c++) {
            int piv = c;
            for (int r = c + 1; r < 6; r++)
                if (fabs(s_H[r][c]) > fabs(s_H[piv][c])) piv = r;
            if (s_H[piv][c] == 0.0) { ok = false; break; }
            if (piv != c)
                for (int j = 0; j < 12; j++) { const double t = s_H[c][j]; s_H[c][j] = s_H[piv][j]; s_H[piv][j] = t; }
            const double d = s_H[c][c];
            for (int j = 0; j < 12; j++) s_H[c][j] /= d;
            for (int r = 0; r < 6; r++) {
                if (r == c) continue;
                const double f = s_H[r][c];
                if (f != 0.0)
                    for (int j = 0; j < 12; j++) s_H[r][j] -= f * s_H[c][j];
            }
        }
        double *out = cov36 + (size_t)link * 36;
        if (!ok) {
            for (int k = 0; k < 36; k++) out[k] = 0.0;
            status[link] = 1;                      // singular Hessian
        } else {
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < 6; b++) {
                    double s2 = 0;
                    for (int k = 0; k < 6; k++) s2 += s_H[a][6 + k] * s_JK[k][b];
                    s_tmp[a][b] = s2;
                }
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < 6; b++) {
                    double s2 = 0;
                    for (int k = 0; k < 6; k++) s2 += s_tmp[a][k] * s_H[k][6 + b];
                    out[a * 6 + b] = s2;
                }
            status[link] = 0;
        }
    }
}

// One evaluation of derivativesNDT for ONE pair spread over many workgroups (host-driven matcher for
// small batches / large maps: a single registration then uses the whole chip instead of one CU).
// Workgroup g evaluates a contiguous share of the source cells and writes its 28 partial sums;
// the host adds the partials in workgroup order.
template <int NN>
__global__ __launch_bounds__(NDT_MATCH_THREADS) void ndt_eval_kernel(
    NdtSetView tset, unsigned tmap, NdtSetView sset, unsigned smap, rigid T, int with_h, double lfd1, double lfd2,
    double *__restrict__ partials)
{
    __shared__ EvalShared<NDT_MATCH_WAVES> sh;
    const MapView tg = map_view(tset, tmap);
    const MapView sv = map_view(sset, smap);
    const int per = (sv.n_cells + (int)gridDim.x - 1) / (int)gridDim.x;
    const int begin = min(sv.n_cells, (int)blockIdx.x * per), count = min(sv.n_cells - begin, per);
    if (threadIdx.x < 32) sh.sums[threadIdx.x] = 0.0;
    __syncthreads();
    if (with_h) eval_derivs<NN, true>(tg, sv.cells + begin, count, T, lfd1, lfd2, sh);
    else eval_derivs<NN, false>(tg, sv.cells + begin, count, T, lfd1, lfd2, sh);
    if (threadIdx.x < 32) partials[blockIdx.x * 32 + threadIdx.x] = sh.sums[threadIdx.x];
}

// ONE registration spread over the whole launch grid (small batches / large maps: the reference's
// one-link-at-a-time call pattern).  Every workgroup evaluates a contiguous share of the source cells and leaves
// 28 partial sums in global memory; after a grid barrier workgroup 0 adds the partials in workgroup order (fixed:
// deterministic) and runs the solver step, publishes the next request, and a second barrier releases everybody.
// Nothing goes back to the host between evaluations.  The launcher keeps gridDim.x <= number of CUs (one 512-thread
// workgroup per CU is resident), the barrier spins with a bound and raises `abort` instead of hanging.
struct NdtCoopCtrl {
    unsigned top, abort, leave, pad1;    // top: groups that completed a barrier, summed over all barriers so far;
                                         // leave: workgroups that are through with the block (the last one zeroes it)
    rigid Teval;
    int with_h, done;
    alignas(64) unsigned grp[16 * 16];   // arrival counter of workgroup group i at grp[16 * i] (one 64-byte line each)
};
static_assert(sizeof(NdtCoopCtrl) % 64 == 0, "control block keeps the partials aligned");
size_t ndt_match_coop_work_bytes(size_t n_chunks) { return sizeof(NdtCoopCtrl) + 2 * n_chunks * 32 * sizeof(double); }   // multiple of 64; two sets of rows (ndt_match_coop_kernel)
size_t ndt_match_coop_ctrl_bytes() { return sizeof(NdtCoopCtrl); }   // what a launch sequence must find zeroed

// Grid barrier number `epoch` (1, 2, ...).  Arrivals on ONE counter serialise at the L2 (~50-100 ns each: 30 us for
// 256 workgroups), so workgroups arrive on one of up to 16 group counters and the last arrival of a group bumps
// the top counter everybody polls: ~16 + 16 serialised atomics instead of 256.  Counters only grow (no reset
// race); nobody can arrive for barrier e + 1 before every workgroup has arrived for barrier e.
NDT_D bool coop_barrier(NdtCoopCtrl *c, unsigned &epoch, unsigned G)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        epoch += 1u;
        const unsigned NG = G < 16u ? G : 16u, gi = blockIdx.x % NG;
        const unsigned gsize = (G - gi + NG - 1u) / NG;            // workgroups w with w % NG == gi
        // (up to 16 workgroups: every group is one workgroup, and its arrival goes straight to the top counter -- one atomic
        //  round trip less per barrier, 18 barriers in the registration of a planar pair)
        if (G <= 16u) atomicAdd(&c->top, 1u);
        else if (atomicAdd(&c->grp[16u * gi], 1u) + 1u == gsize * epoch) atomicAdd(&c->top, 1u);
        unsigned spins = 0;
        while (__hip_atomic_load(&c->top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NG * epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 24)) { __hip_atomic_store(&c->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            if ((spins & 1023u) == 0u && __hip_atomic_load(&c->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        }
    }
    __syncthreads();
    return __hip_atomic_load(&c->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
}

// Thread (r, k) of 16 x 32 adds value k of rows r, r + 16, ... of `rows` (NC rows of 32 doubles written by other
// workgroups: system-scope loads).  The loads of a thread are issued EIGHT at a time before the first is used: written as
// `a += load` the compiler waits for every load before it issues the next (they are atomic accesses and stay in program
// order), and a registration of 96 chunks paid six dependent round trips across the fabric per evaluation.  Same order
// of additions as the plain loop.
static __device__ __noinline__ double sum_rows_16x32(const double *rows, unsigned NC, unsigned r, unsigned k)
{
    auto ldd = [](const double *q) {
        return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    };
    double a = 0;
#pragma unroll 1
    for (unsigned w0 = r; w0 < NC; w0 += 128u) {
        double v[8];
#pragma unroll
        for (unsigned j = 0; j < 8u; j++) v[j] = ldd(rows + (size_t)min(w0 + 16u * j, NC - 1u) * 32 + k);
#pragma unroll
        for (unsigned j = 0; j < 8u; j++) if (w0 + 16u * j < NC) a += v[j];
    }
    return a;
}

template <int NN>
__global__ __launch_bounds__(NDT_MATCH_THREADS) void ndt_match_coop_kernel(
    NdtSetView tset, const uint32_t *__restrict__ tidx, NdtSetView sset, const uint32_t *__restrict__ sidx,
    double *__restrict__ T16_all, NdtMatchParamsDev prm, NdtMatchResultDev *__restrict__ res_all,
    const double *__restrict__ Q36_all, char *__restrict__ work_all, size_t work_stride, unsigned cells_per_group,
    unsigned pair_begin, unsigned *__restrict__ done_host /* NULL, or one word per pair in host memory: set when pose and result are written */)
{
    __shared__ EvalShared<NDT_MATCH_WAVES> sh;
    __shared__ MatchState st;       // every workgroup its own copy, all alike (see below)
    __shared__ NewtonWs s_ws;
    __shared__ NdtMatchParamsDev s_prm;     // the solver takes the parameters by reference: LDS, not a private copy
    __shared__ long long s_cnt[5];          // clocks: evaluations, solver, barriers; pair terms: gradient-only, with Hessian
    __shared__ double s_rows[4 * NDT_VW * 32];      // eval_chunks: up to 4 chunks x 8 shares x 32 sums
    __shared__ double s_out[4 * 32];                // ... and their sums over the shares

    // blockIdx.y = registration: its gridDim.x workgroups have their own control block and barrier.  The caller sizes
    // the grid by the occupancy query and lets one such launch run at a time: every workgroup that stays is resident,
    // so no barrier can wait for a workgroup that has not started.
    // What the workgroups of a registration hand to each other (the rows of partial sums) is stored
    // write-through and LOADED with system-scope accesses (past the L1 and the XCD's L2); a writer's stores are complete
    // (vmcnt 0) before its workgroup arrives at the barrier.  The barrier itself is agent-scope atomics.  No
    // cache-maintenance fence (1.7 - 3.5 us each, two or three per barrier before): the maps stay in L1 / L2.
    auto st64 = [](void *q, unsigned long long v) { __hip_atomic_store(reinterpret_cast<unsigned long long *>(q), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    auto std_ = [&](double *q, double v) { st64(q, __builtin_bit_cast(unsigned long long, v)); };
    const unsigned pair = pair_begin + blockIdx.y;
    char *work_mem = work_all + (size_t)pair * work_stride;
    double *T16 = T16_all + (size_t)pair * 16;
    NdtMatchResultDev *res = res_all + pair;
    const double *Q36 = Q36_all ? Q36_all + (size_t)pair * 36 : nullptr;
    NdtCoopCtrl *ctrl = reinterpret_cast<NdtCoopCtrl *>(work_mem);
    double *partials = reinterpret_cast<double *>(work_mem + sizeof(NdtCoopCtrl));
    // the indices and the maps may come from device memory the host never saw: check them here (like the persistent kernel)
    const uint32_t ti = tidx[pair], si = sidx[pair];
    const bool bad_index = ti >= tset.n_maps || si >= sset.n_maps;
    const bool truncated = !bad_index && (tset.counters[ti].overflow != 0u || sset.counters[si].overflow != 0u);
    if (bad_index || truncated) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {              // (converged = 0; the pose is left untouched)
            NdtMatchResultDev o;
            o.converged = 0; o.iterations = 0; o.fevals = 0;
            o.exit_code = bad_index ? -2 : -3;                  // -2 map index out of range, -3 a map needed more cells than max_cells
            o.score = 0.0; o.n_source = 0; o.n_target = 0;
            o.cycles_eval = 0; o.cycles_solver = 0; o.pair_terms_g = 0; o.pair_terms_h = 0;
            *res = o;
            if (done_host) { __threadfence_system(); __hip_atomic_store(&done_host[pair], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
        return;
    }
    const MapView tg = map_view_uniform(tset, ti);
    const MapView sv = map_view_uniform(sset, si);
    // The source cells are cut into NC CHUNKS of (about) cells_per_group cells -- a property of the map alone -- and every
    // chunk has its own row of partial sums; the rows are added in chunk order.  The G workgroups that the grid has for
    // this registration deal the chunks among themselves (workgroup g: chunks g, g + G, ...): one chunk each when a
    // single registration has the chip, a dozen each when 32 registrations of 12 k cells share it.  The result does not
    // depend on G, i.e. not on the batch a registration is in.  Surplus workgroups leave at once.
    const unsigned g = blockIdx.x;
    const unsigned NC = max(1u, ((unsigned)sv.n_cells + cells_per_group - 1u) / cells_per_group);
    // chunks of equal length, a multiple of 8 cells (a chunk is then a whole number of lanes in each of the 8 shares)
    const int per = (((sv.n_cells + (int)NC - 1) / (int)NC) + 7) & ~7;
    // fewer workgroups than chunks: a workgroup takes CH chunks at a time through ONE pass of up to 64 cells per wave
    // (eval_chunks; with 16 cells per wave TRANSFORM and PROBE would run at a quarter of their width)
    const unsigned seg_lanes = (unsigned)per / 8u;
    // (as few chunks per pass as give every workgroup at most one pass: 4 registrations of 96 chunks on 64 workgroups each
    //  take TWO chunks per pass on 48 workgroups -- four per pass kept 24 busy: 2.03 against 1.34 ms)
    const bool packed = gridDim.x < NC && seg_lanes >= 8u && seg_lanes <= 32u;
    const unsigned CH = !packed ? 1u : min(64u / seg_lanes, NC <= 2u * gridDim.x ? 2u : 4u);
    const unsigned NQ = (NC + CH - 1u) / CH;                     // passes of CH chunks
    const unsigned G = min(gridDim.x, NQ);
    if (g >= G) return;
    unsigned target = 0;
    // (profiling counters of workgroup 0 in LDS: as registers they are live across the whole loop and get spilled)
    if (threadIdx.x < 5) s_cnt[threadIdx.x] = 0;

    if (threadIdx.x == 0) s_prm = prm;
    __syncthreads();
    // EVERY workgroup keeps the solver state and steps it itself (round 6).  The rows of partial sums are what the workgroups
    // hand to each other; the sums of the rows, in chunk order, are the same doubles in every workgroup, the solver step is the
    // same code on the same inputs, so the states never differ -- and nobody has to publish the next pose: ONE grid barrier per
    // evaluation instead of two, and no request to fetch across the fabric (a 2D pair of 100 k points: 0.32 -> see DESIGN 7).
    // The rows of consecutive evaluations go to alternating halves of the row buffer: a workgroup that is already writing the
    // rows of evaluation e + 1 cannot disturb one that still reads those of e, and nobody writes the rows of e + 2 before
    // everybody has arrived at the barrier of e + 1, i.e. has read the rows of e.
    if (threadIdx.x == 0) match_state_init(st, T16, s_prm, Q36);
    __syncthreads();
    // a barrier that gave up (a foreign process holding CUs: bounded spin): the registration reports exit code -4 and
    // leaves the pose as it came in; the host-pointer entries run it again on the persistent kernel
    auto gave_up = [&]() {
        if (g == 0 && threadIdx.x == 0) {
            NdtMatchResultDev o;
            o.converged = 0; o.iterations = 0; o.fevals = 0; o.exit_code = -4;
            o.score = 0.0; o.n_source = sv.n_cells; o.n_target = tg.n_cells;
            o.cycles_eval = 0; o.cycles_solver = 0; o.pair_terms_g = 0; o.pair_terms_h = 0;
            *res = o;
            if (done_host) { __threadfence_system(); __hip_atomic_store(&done_host[pair], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
    };
    unsigned parity = 0u;
    for (;;) {
        if (st.done) break;               // (LDS, written before the last barrier of the workgroup's threads: the same everywhere)
        const rigid &Te = st.Teval;       // (read from LDS where the transform needs it, see ndt_match_pool_kernel)
        const int with_h = st.with_h;
        double *rows = partials + (size_t)parity * NC * 32;
        long long c0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (unsigned q = g; q < NQ; q += G) {
            const unsigned c = q * CH;
            const int begin = min(sv.n_cells, (int)c * per), count = min(sv.n_cells - begin, (int)CH * per);
            if (packed) {
                const unsigned n_out = min(CH, NC - c);
                if (with_h) eval_chunks<NN, true>(tg, sv.cells + begin, count, Te, prm.lfd1, prm.lfd2, sh, seg_lanes, s_rows, s_out, n_out);
                else eval_chunks<NN, false>(tg, sv.cells + begin, count, Te, prm.lfd1, prm.lfd2, sh, seg_lanes, s_rows, s_out, n_out);
                if (threadIdx.x < n_out * 32u) std_(rows + c * 32 + threadIdx.x, s_out[threadIdx.x]);
            } else {
                if (with_h) eval_derivs<NN, true>(tg, sv.cells + begin, count, Te, prm.lfd1, prm.lfd2, sh);
                else eval_derivs<NN, false>(tg, sv.cells + begin, count, Te, prm.lfd1, prm.lfd2, sh);
                if (threadIdx.x < 32) std_(rows + c * 32 + threadIdx.x, sh.sums[threadIdx.x]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the rows are in memory before the workgroup arrives
        long long c1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) s_cnt[0] += c1 - c0;
        if (!coop_barrier(ctrl, target, G)) { gave_up(); return; }   // all rows of this evaluation are in memory
        if (threadIdx.x == 0) s_cnt[2] += (long long)__builtin_readcyclecounter() - c1;
        // 16 x 32 threads: thread (r, k) adds value k of chunks r, r + 16, ... (loads of different threads
        // overlap: a single lane walking all NC rows pays NC dependent L2 round trips); the 16 rows are then
        // added in order.  Fixed order: deterministic, and the same in every workgroup.
        {
            unsigned t = threadIdx.x;
            asm volatile("" : "+v"(t));           // (recomputed here: hoisted out of the loop the shift is kept in a spilled register)
            const unsigned k = t & 31u, r = t >> 5;
            sh.src[r * 32 + k] = sum_rows_16x32(rows, NC, r, k);    // the source tile buffer is free between evaluations
        }
        __syncthreads();
        if (threadIdx.x < 29) {
            double a = 0;
            for (unsigned r = 0; r < 16u; r++) a += sh.src[r * 32 + threadIdx.x];
            sh.sums[threadIdx.x] = a;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            long long d0 = __builtin_readcyclecounter();
            s_cnt[with_h ? 4 : 3] += (long long)sh.sums[28];
            match_state_step(st, sh.sums, s_prm, s_ws);
            s_cnt[1] += (long long)__builtin_readcyclecounter() - d0;
        }
        __syncthreads();
        parity ^= 1u;
    }
    // The barrier counters only grow while a registration runs and must be zero when the next one starts: the last
    // workgroup to leave (every other one has read the block for the last time) puts them back, so that the host does
    // not have to clear the block before every call (a registration that gave up leaves it dirty; the host clears then).
    if (threadIdx.x == 0) {
        if (atomicAdd(&ctrl->leave, 1u) + 1u == G) {
            const unsigned NG = G < 16u ? G : 16u;
            for (unsigned i = 0; i < NG; i++) __hip_atomic_store(&ctrl->grp[16u * i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctrl->top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctrl->leave, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (g == 0 && threadIdx.x == 0) {
        NdtMatchResultDev o;
        match_state_result(st, T16, o);
        o.n_source = sv.n_cells;
        o.n_target = tg.n_cells;
        o.cycles_eval = s_cnt[0];          // workgroup 0: its share of the evaluations
        o.cycles_solver = s_cnt[1];
#ifdef NDT_COOP_PROF
        o.cycles_eval = s_cnt[2];          // profiling build: time at the grid barriers instead
#endif
        o.pair_terms_g = s_cnt[3];
        o.pair_terms_h = s_cnt[4];
        *res = o;
        // (a host that polls this word instead of waiting for the stream: pose and result are visible before it is)
        if (done_host) { __threadfence_system(); __hip_atomic_store(&done_host[pair], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
}

// Workgroups of the cooperative matcher that can be resident at once (occupancy query x CUs): the grid barrier of a
// launch is only safe when the whole grid is co-resident.
unsigned ndt_match_coop_capacity(int n_neighbours)
{
    int dev = 0, n_cu = 0, per_cu = 0;
    if (n_neighbours < 0 || n_neighbours > 3 || hipGetDevice(&dev) != hipSuccess) return 0;
    static unsigned cache[16][4];      // the query costs tens of microseconds: once per device and kernel
    if (dev >= 0 && dev < 16 && cache[dev][n_neighbours]) return cache[dev][n_neighbours];
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) return 0;
    hipError_t e;
    switch (n_neighbours) {
    case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ndt_match_coop_kernel<0>, NDT_MATCH_THREADS, 0); break;
    case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ndt_match_coop_kernel<1>, NDT_MATCH_THREADS, 0); break;
    case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ndt_match_coop_kernel<2>, NDT_MATCH_THREADS, 0); break;
    case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ndt_match_coop_kernel<3>, NDT_MATCH_THREADS, 0); break;
    default: return 0;
    }
    if (e != hipSuccess || per_cu <= 0) return 0;
    if (per_cu > 1) per_cu = 1;     // 85 KB of LDS and 8 x 256 VGPRs: one workgroup per CU, whatever the query says at an edge
    if (dev >= 0 && dev < 16) cache[dev][n_neighbours] = (unsigned)(per_cu * n_cu);
    return (unsigned)(per_cu * n_cu);
}

// ONE launch of the grid-barrier matcher for pairs [pair_begin, pair_begin + pair_count); the control blocks are zero.
// The caller sizes the grid by the occupancy query (every workgroup that stays is resident) and lets one such launch
// run at a time per process; with that a plain launch is as safe as hipLaunchCooperativeKernel, which adds 0.08 ms to
// a call -- more for every other stream the process has (it goes through a queue of its own that is ordered against
// all of them: 0.35 -> 0.43 ms with three idle streams, 0.58 ms inside bench.py; tools/latency_probe.py).  What the
// runtime's check would add -- a foreign process holding CUs -- is covered by the barrier's bounded spin and the
// caller's re-run.  `checked` != 0 (NDTGPU_COOP_API=1) uses the cooperative API.
hipError_t ndt_launch_match_coop(const NdtSetView &tset, const uint32_t *tidx_dev, const NdtSetView &sset,
                                 const uint32_t *sidx_dev, double *T16_dev, size_t pair_begin, size_t pair_count,
                                 const NdtMatchParamsDev &prm, NdtMatchResultDev *res_dev, const double *Q36_dev,
                                 unsigned n_groups, unsigned cells_per_group, void *work_dev, size_t work_stride, int checked,
                                 hipStream_t stream, unsigned *done_host)
{
    size_t stride = work_stride;        // per registration: a control block + one row of partial sums per chunk
    NdtSetView ts = tset, ss = sset;
    NdtMatchParamsDev p = prm;
    char *work = (char *)work_dev;
    unsigned pb = (unsigned)pair_begin;
    void *args[] = {&ts, &tidx_dev, &ss, &sidx_dev, &T16_dev, &p, &res_dev, &Q36_dev, &work, &stride, &cells_per_group, &pb, &done_host};
    const dim3 grid(n_groups, (unsigned)pair_count), block(NDT_MATCH_THREADS);
    if (!checked) {
#define NDT_LAUNCH_COOP(NN)                                                                                            \
    hipLaunchKernelGGL(ndt_match_coop_kernel<NN>, grid, block, 0, stream, ts, tidx_dev, ss, sidx_dev, T16_dev, p, res_dev, \
                       Q36_dev, work, stride, cells_per_group, pb, done_host)
        switch (prm.n_neighbours) {
        case 0: NDT_LAUNCH_COOP(0); break;
        case 1: NDT_LAUNCH_COOP(1); break;
        case 2: NDT_LAUNCH_COOP(2); break;
        case 3: NDT_LAUNCH_COOP(3); break;
        default: return hipErrorInvalidValue;
        }
#undef NDT_LAUNCH_COOP
        return hipGetLastError();
    }
    switch (prm.n_neighbours) {
    case 0: return hipLaunchCooperativeKernel((const void *)ndt_match_coop_kernel<0>, grid, block, args, 0, stream);
    case 1: return hipLaunchCooperativeKernel((const void *)ndt_match_coop_kernel<1>, grid, block, args, 0, stream);
    case 2: return hipLaunchCooperativeKernel((const void *)ndt_match_coop_kernel<2>, grid, block, args, 0, stream);
    case 3: return hipLaunchCooperativeKernel((const void *)ndt_match_coop_kernel<3>, grid, block, args, 0, stream);
    default: return hipErrorInvalidValue;
    }
}

// ---- the task-pool matcher: batches that cannot fill the chip with one workgroup per registration ---------------------
// One EVALUATION of one registration is a set of TASKS (one chunk of ~128 source cells each, or four chunks through one
// pass of 64 cells per wave while the registrations still running have more chunks than the launch has workgroups -- decided
// anew for every evaluation, so the last registrations of a batch are cut fine and spread over the idle chip); any workgroup
// takes any open task of any registration (a 64-bit ticket per registration: evaluation | tasks | tasks drawn), writes the
// chunks' rows of partial
// sums and counts the task as delivered.  The workgroup that delivers the LAST task of an evaluation adds the rows in chunk
// order, runs the solver step (state in global memory between steps, ~1.5 KB) and publishes the next evaluation, or the
// result.  No barrier between workgroups and nothing that needs them resident together: a launch cannot deadlock, needs no
// occupancy-sized grid and no ordering against other launches; registrations that run long get the workgroups the others
// leave.  Chunks are a property of the map and rows are added in chunk order: a registration's result does not depend on
// who computed what, nor on the batch -- bit for bit.
struct alignas(64) NdtPoolPair {
    unsigned long long ticket;          // evaluation << 40 | tasks of it << 20 | tasks drawn; evaluation 0: nothing published
    unsigned n_tasks, done_tasks;       // (unused); tasks delivered in the open evaluation
    int with_h;
    unsigned chunks_per_task, n_chunks, per;   // the cut of this registration's source cells
    rigid Teval;                        // pose of the open evaluation
    long long cnt[4];                   // clocks in evaluations (all workgroups) and solver steps; pair terms g / h
    MatchState st;                      // solver state between steps
};
struct alignas(64) NdtPoolCtrl {
    unsigned finished, abort, leave, pad;
};
size_t ndt_match_pool_ctrl_bytes() { return sizeof(NdtPoolCtrl); }
size_t ndt_match_pool_head_bytes() { return sizeof(NdtPoolPair); }          // per registration: what a launch must find zeroed
size_t ndt_match_pool_pair_bytes(size_t n_chunks) { return sizeof(NdtPoolPair) + n_chunks * 32 * sizeof(double); }

template <int NN>
__global__ __launch_bounds__(NDT_MATCH_THREADS) void ndt_match_pool_kernel(
    NdtSetView tset, const uint32_t *__restrict__ tidx, NdtSetView sset, const uint32_t *__restrict__ sidx,
    double *__restrict__ T16_all, NdtMatchParamsDev prm, NdtMatchResultDev *__restrict__ res_all,
    const double *__restrict__ Q36_all, char *__restrict__ work, size_t pair_stride, unsigned cells_per_group,
    unsigned n_pairs)
{
    __shared__ EvalShared<NDT_MATCH_WAVES> sh;
    __shared__ MatchState st;
    __shared__ NewtonWs s_ws;
    __shared__ NdtMatchParamsDev s_prm;     // the solver takes the parameters by reference: LDS, not a private copy
    __shared__ rigid s_T;
    __shared__ double s_rows[4 * NDT_VW * 32];      // eval_chunks: up to 4 chunks x 8 shares x 32 sums
    __shared__ double s_out[4 * 32];                // ... and their sums over the shares
    __shared__ unsigned s_task[12];         // pair, task, with_h, code, last, evaluation, chunks, chunks per task, cells per chunk, tasks, next chunks per task
    __shared__ long long s_clk;
    enum { POOL_TASK = 0, POOL_NONE = 1, POOL_EXIT = 2 };
    const unsigned tid = threadIdx.x;
    NdtPoolCtrl *ctl = reinterpret_cast<NdtPoolCtrl *>(work);
    auto pair_at = [&](unsigned p) { return reinterpret_cast<NdtPoolPair *>(work + sizeof(NdtPoolCtrl) + (size_t)p * pair_stride); };
    auto rows_of = [&](unsigned p) { return reinterpret_cast<double *>(work + sizeof(NdtPoolCtrl) + (size_t)p * pair_stride + sizeof(NdtPoolPair)); };
    auto aload = [](const unsigned *q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto tk_make = [](unsigned seq, unsigned nt) { return ((unsigned long long)seq << 40) | ((unsigned long long)nt << 20); };
    auto tk_seq = [](unsigned long long t) { return (unsigned)(t >> 40); };
    auto tk_tasks = [](unsigned long long t) { return (unsigned)(t >> 20) & 0xFFFFFu; };
    auto tk_drawn = [](unsigned long long t) { return (unsigned)t & 0xFFFFFu; };
    // chunks per task of the NEXT evaluation of a registration of NC chunks (seg_lanes lanes per chunk and share): four
    // through one pass while the registrations still running have more chunks than the launch has workgroups, one otherwise
    auto chunks_per_task = [&](unsigned NC, unsigned seg_lanes) -> unsigned {
        const unsigned active = n_pairs - min(n_pairs, aload(&ctl->finished));
        // (two chunks per task in the tail as well -- half as many draws and deliveries per evaluation -- was measured: 32
        //  pairs of 12 k-cell maps 4.90 against 4.75 ms, four: 5.00)
        const bool packed = (size_t)active * NC > gridDim.x && seg_lanes >= 8u && seg_lanes <= 32u;
        return packed ? min(4u, 64u / seg_lanes) : 1u;
    };
    // Everything workgroups hand to each other (request, rows, solver state) is LOADED with system-scope accesses, which
    // bypass the L1 and the XCD's L2, and stored write-through; a producer's stores are complete (vmcnt 0, at the
    // workgroup barrier) before its thread 0 touches the ticket / counter the consumer polls.  No cache-maintenance fence
    // anywhere (1.7 - 3.5 us each, MI355X_MICROARCH.md): the maps stay in L1 / L2 from task to task.
    auto ld64 = [](const void *q) { return __hip_atomic_load(reinterpret_cast<const unsigned long long *>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    auto st64 = [](void *q, unsigned long long v) { __hip_atomic_store(reinterpret_cast<unsigned long long *>(q), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    auto ld32 = [](const void *q) { return __hip_atomic_load(reinterpret_cast<const unsigned *>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    auto st32 = [](void *q, unsigned v) { __hip_atomic_store(reinterpret_cast<unsigned *>(q), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    auto ldd = [&](const double *q) { return __builtin_bit_cast(double, ld64(q)); };
    auto std_ = [&](double *q, double v) { st64(q, __builtin_bit_cast(unsigned long long, v)); };
    constexpr unsigned ST_WORDS = (unsigned)(sizeof(MatchState) / sizeof(unsigned long long));
    static_assert(sizeof(MatchState) % sizeof(unsigned long long) == 0, "MatchState is copied as 64-bit words");
    auto refuse = [&](unsigned pair, int code) {                // (converged = 0; the pose is left untouched)
        NdtMatchResultDev *o = res_all + pair;
        o->converged = 0; o->iterations = 0; o->fevals = 0; o->exit_code = code;
        o->score = 0.0; o->n_source = 0; o->n_target = 0;
        o->cycles_eval = 0; o->cycles_solver = 0; o->pair_terms_g = 0; o->pair_terms_h = 0;
    };
    auto finish_pair = [&](NdtPoolPair *P, unsigned pair) {     // thread 0, solver state in `st`
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (this thread's own counter updates are in memory before it reads them)
        NdtMatchResultDev o;
        match_state_result(st, T16_all + (size_t)pair * 16, o);
        o.n_source = (int)sset.counters[sidx[pair]].n_cells;
        o.n_target = (int)tset.counters[tidx[pair]].n_cells;
        o.cycles_eval = (long long)ld64(&P->cnt[0]); o.cycles_solver = (long long)ld64(&P->cnt[1]);
        o.pair_terms_g = (long long)ld64(&P->cnt[2]); o.pair_terms_h = (long long)ld64(&P->cnt[3]);
        res_all[pair] = o;
        __threadfence();                // (once per registration: pose and result leave this XCD's L2 before the launch can end)
        atomicAdd(&ctl->finished, 1u);
    };
    if (tid == 0) s_prm = prm;
    __syncthreads();

    // ---- open the registrations (pair p by workgroup p mod gridDim.x) -------------------------------------------------
    for (unsigned pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
        if (tid == 0) {
            NdtPoolPair *P = pair_at(pair);
            // the indices and the maps may come from device memory the host never saw: check them here
            const uint32_t ti = tidx[pair], si = sidx[pair];
            const bool bad_index = ti >= tset.n_maps || si >= sset.n_maps;
            const bool truncated = !bad_index && (tset.counters[ti].overflow != 0u || sset.counters[si].overflow != 0u);
            if (bad_index || truncated) {
                refuse(pair, bad_index ? -2 : -3);          // -2 map index out of range, -3 a map needed more cells than max_cells
                atomicAdd(&ctl->finished, 1u);
            } else {
                refuse(pair, -4);                           // (what stays if the launch gives up before this one is finished)
                __threadfence();                            // (... written through before any other workgroup can finish the pair)
                match_state_init(st, T16_all + (size_t)pair * 16, s_prm, Q36_all ? Q36_all + (size_t)pair * 36 : nullptr);
                if (st.done) {                               // parameters the solver rejects
                    for (int i = 0; i < 4; i++) st64(&P->cnt[i], 0ull);
                    finish_pair(P, pair);
                } else {
                    // chunks of equal length, a multiple of 8 cells (a chunk is then a whole number of lanes in each of the
                    // 8 shares); four of them per task when the batch has more tasks than the launch has workgroups
                    const unsigned n = sset.counters[si].n_cells;
                    const unsigned NC = max(1u, (n + cells_per_group - 1u) / cells_per_group);
                    const unsigned per = (((n + NC - 1u) / NC) + 7u) & ~7u;
                    const unsigned seg_lanes = per / 8u;
                    const unsigned CH = chunks_per_task(NC, seg_lanes);
                    st32(&P->n_chunks, NC); st32(&P->per, per); st32(&P->chunks_per_task, CH);
                    st32(&P->done_tasks, 0u);
                    st32(&P->with_h, (unsigned)st.with_h);
                    for (int i = 0; i < 9; i++) std_(&P->Teval.r[i], st.Teval.r[i]);
                    for (int i = 0; i < 3; i++) std_(&P->Teval.t[i], st.Teval.t[i]);
                    for (int i = 0; i < 4; i++) st64(&P->cnt[i], 0ull);
                    for (unsigned i = 0; i < ST_WORDS; i++)
                        st64(reinterpret_cast<unsigned long long *>(&P->st) + i, reinterpret_cast<const unsigned long long *>(&st)[i]);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(&P->ticket, tk_make(1u, (NC + CH - 1u) / CH), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __syncthreads();
    }

    // ---- take tasks until every registration is finished --------------------------------------------------------------
    unsigned home = blockIdx.x % n_pairs, idle = 0;
    for (;;) {
        if (tid < 64) {
            // Wave 0 looks for a task: lane l reads the ticket of registration home + l (one round trip for 64
            // registrations; one lane walking them paid one round trip EACH, and an idle workgroup took 20 us to notice the
            // evaluation that the last registration of 32 had just opened), the first lane with an open ticket draws.
            unsigned code = POOL_NONE;
            // (the two words that end the launch travel with the tickets: read on their own when nothing is open they made
            //  an idle workgroup's look-around three round trips long)
            unsigned ctl_word = 0u;
            if (tid < 2) ctl_word = aload(tid == 0 ? &ctl->finished : &ctl->abort);
            for (unsigned base = 0; base < n_pairs && code == POOL_NONE; base += 64u) {
                const unsigned k = base + tid;
                const unsigned q = (home + k) % n_pairs;
                NdtPoolPair *P = pair_at(q);
                unsigned long long t = 0ull;
                if (k < n_pairs) t = __hip_atomic_load(&P->ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long open = ndt_ballot(tk_seq(t) != 0u && tk_drawn(t) < tk_tasks(t));
                while (open != 0ull && code == POOL_NONE) {
                    const unsigned l = (unsigned)__builtin_ctzll(open);
                    open &= open - 1ull;
                    // Lane l draws; the sixteen lanes after it read the request of that registration AT THE SAME TIME (one
                    // word each).  The ticket was seen open for evaluation S before these loads were issued, so they return
                    // the request of S or of a later evaluation -- and if the draw wins a task of S, S is still open when
                    // it does (this task is undelivered) and its request has not been overwritten.  A draw that wins a task of a
                    // LATER evaluation (S ended in between) may have read a request in the middle of being rewritten: it reads
                    // again, the slow way.
                    NdtPoolPair *Pl = pair_at((home + base + l) % n_pairs);
                    const unsigned j = (tid - l - 1u) & 63u;
                    unsigned long long req64 = 0ull;
                    unsigned req32 = 0u;
                    int won = 0;
                    if (tid == l) {
                        const unsigned seen = tk_seq(t);
                        t = __hip_atomic_fetch_add(&P->ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (tk_seq(t) != 0u && tk_drawn(t) < tk_tasks(t)) {
                            won = 1;
                            s_task[0] = q; s_task[1] = tk_drawn(t); s_task[5] = tk_seq(t); s_task[9] = tk_tasks(t);
                            s_task[11] = tk_seq(t) != seen ? 1u : 0u;
                        }
                    } else if (j < 9u) req64 = ld64(&Pl->Teval.r[j]);
                    else if (j < 12u) req64 = ld64(&Pl->Teval.t[j - 9u]);
                    else if (j == 12u) req32 = ld32(&Pl->with_h);
                    else if (j == 13u) req32 = ld32(&Pl->n_chunks);
                    else if (j == 14u) req32 = ld32(&Pl->chunks_per_task);
                    else if (j == 15u) req32 = ld32(&Pl->per);
                    if (ndt_ballot(won)) {
                        code = POOL_TASK;
                        if (j < 9u) s_T.r[j] = __builtin_bit_cast(double, req64);
                        else if (j < 12u) s_T.t[j - 9u] = __builtin_bit_cast(double, req64);
                        else if (j == 12u) s_task[2] = req32;
                        else if (j == 13u) s_task[6] = req32;
                        else if (j == 14u) s_task[7] = req32;
                        else if (j == 15u) s_task[8] = req32;
                    }
                }
            }
            if (code == POOL_TASK) {
                ndt_wave_sync();
                if (s_task[11] != 0u) {
                    NdtPoolPair *P = pair_at(s_task[0]);
                    if (tid < 9) s_T.r[tid] = ldd(&P->Teval.r[tid]);
                    else if (tid < 12) s_T.t[tid - 9] = ldd(&P->Teval.t[tid - 9]);
                    else if (tid == 12) s_task[2] = ld32(&P->with_h);
                    else if (tid == 13) s_task[6] = ld32(&P->n_chunks);
                    else if (tid == 14) s_task[7] = ld32(&P->chunks_per_task);
                    else if (tid == 15) s_task[8] = ld32(&P->per);
                }
                home = s_task[0];                                             // (look here first next time)
                idle = 0;
            } else {
                const unsigned long long ended = ndt_ballot((tid == 0 && ctl_word >= n_pairs) || (tid == 1 && ctl_word != 0u));
                if (ended != 0ull) code = POOL_EXIT;
                // nothing open: somebody is in a solver step (or everything left is being evaluated).  ~10 s of this
                // means a bug, not a wait: raise the abort word instead of hanging the device
                else if (++idle > (1u << 23)) {
                    if (tid == 0) __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    code = POOL_EXIT;
                }
            }
            if (tid == 0) s_task[3] = code;
        }
        __syncthreads();
        const unsigned code = s_task[3];
        if (code == POOL_EXIT) break;
        if (code == POOL_NONE) { __builtin_amdgcn_s_sleep(8); __syncthreads(); continue; }
        // (what a task needs lives in LDS and is read again after the evaluation: as registers across it, it is spilled)
        unsigned tt = threadIdx.x;
        asm volatile("" : "+v"(tt));
        if (tt == 0) s_clk = (long long)__builtin_readcyclecounter();
        {
            const unsigned pair = s_task[0], task = s_task[1];
            const bool with_h = s_task[2] != 0u;
            double *rows = rows_of(pair);
            const MapView tg = map_view_uniform(tset, tidx[pair]);
            const MapView sv = map_view_uniform(sset, sidx[pair]);
            const unsigned NC = s_task[6], CH = s_task[7];
            const int per = (int)s_task[8];
            const rigid &Te = s_T;        // (read from LDS where the transform needs it: as a register copy it lived across the pair terms, partly in scratch memory)
            const unsigned c = task * CH;
            const int begin = min(sv.n_cells, (int)c * per), count = min(sv.n_cells - begin, (int)CH * per);
            if (CH > 1u) {
                const unsigned seg_lanes = (unsigned)per / 8u;
                const unsigned n_out = min(CH, NC - c);
                if (with_h) eval_chunks<NN, true>(tg, sv.cells + begin, count, Te, prm.lfd1, prm.lfd2, sh, seg_lanes, s_rows, s_out, n_out);
                else eval_chunks<NN, false>(tg, sv.cells + begin, count, Te, prm.lfd1, prm.lfd2, sh, seg_lanes, s_rows, s_out, n_out);
                if (tt < n_out * 32u) std_(rows + c * 32 + tt, s_out[tt]);
            } else {
                if (with_h) eval_derivs<NN, true>(tg, sv.cells + begin, count, Te, prm.lfd1, prm.lfd2, sh);
                else eval_derivs<NN, false>(tg, sv.cells + begin, count, Te, prm.lfd1, prm.lfd2, sh);
                if (tt < 32) std_(rows + c * 32 + tt, sh.sums[tt]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the rows are in memory before the task counts as delivered
        }
        __syncthreads();
        const unsigned pair = s_task[0];
        const bool with_h = s_task[2] != 0u;
        NdtPoolPair *P = pair_at(pair);
        double *rows = rows_of(pair);
        const unsigned NC = s_task[6], n_tasks = s_task[9];
        if (tt == 0) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&P->cnt[0]), (unsigned long long)((long long)__builtin_readcyclecounter() - s_clk));
            const unsigned d = __hip_atomic_fetch_add(&P->done_tasks, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_task[4] = d + 1u == n_tasks ? 1u : 0u;
        }
        __syncthreads();
        if (s_task[4] == 0u) continue;
        // ---- the last task of the evaluation: add the rows, run the solver step, publish ---------------------------------
        // (the thread index is read again here, opaquely: what is derived from it -- LDS addresses, row and word indices --
        //  would otherwise be hoisted out of the loop into registers that live across the evaluations, and be spilled)
        unsigned ft = threadIdx.x;
        asm volatile("" : "+v"(ft));
        {
            // 16 x 32 threads: thread (r, k) adds value k of chunks r, r + 16, ...; the 16 rows are then added in order
            const unsigned k = ft & 31u, r = ft >> 5;
            sh.src[r * 32 + k] = sum_rows_16x32(rows, NC, r, k);       // the source tile buffer is free between evaluations
        }
        {
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&P->st);
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(&st);
            for (unsigned i = ft; i < ST_WORDS; i += NDT_MATCH_THREADS) dst[i] = ld64(src + i);
        }
        __syncthreads();
        if (ft < 29) {
            double a = 0;
            for (unsigned r = 0; r < 16u; r++) a += sh.src[r * 32 + ft];
            sh.sums[ft] = a;
        }
        __syncthreads();
        // (how the NEXT evaluation is cut depends on how many registrations are still running: that word is read by another
        //  wave while thread 0 runs the solver step, and travels with the request)
        if (ft == 64) s_task[10] = chunks_per_task(NC, s_task[8] / 8u);
        if (ft == 0) {
            // (the delivery counter goes back to zero now: nobody touches it before the next ticket, and the store is long
            //  complete when that is published after the solver step)
            __hip_atomic_store(&P->done_tasks, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long d0 = __builtin_readcyclecounter();
            // (counters: atomics that return nothing -- read, add, store cost two round trips across the fabric per evaluation)
            atomicAdd(reinterpret_cast<unsigned long long *>(&P->cnt[with_h ? 3 : 2]), (unsigned long long)(long long)sh.sums[28]);
            match_state_step(st, sh.sums, s_prm, s_ws);
            atomicAdd(reinterpret_cast<unsigned long long *>(&P->cnt[1]), (unsigned long long)((long long)__builtin_readcyclecounter() - d0));
        }
        __syncthreads();
        if (st.done) {
            if (ft == 0) finish_pair(P, pair);
        } else {
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(&P->st);
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&st);
            for (unsigned i = ft; i < ST_WORDS; i += NDT_MATCH_THREADS) st64(dst + i, src[i]);
            if (ft < 9) std_(&P->Teval.r[ft], st.Teval.r[ft]);
            else if (ft < 12) std_(&P->Teval.t[ft - 9], st.Teval.t[ft - 9]);
            else if (ft == 12) st32(&P->with_h, (unsigned)st.with_h);
            else if (ft == 13) st32(&P->chunks_per_task, s_task[10]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (ft == 0) {
                const unsigned CHn = s_task[10];
                __hip_atomic_store(&P->ticket, tk_make(s_task[5] + 1u, (NC + CHn - 1u) / CHn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
    }
    // The tickets must read "nothing published" when the next launch starts: the last workgroup to leave puts the
    // headers back (a launch that gave up leaves them dirty; the host clears then).
    if (tid == 0 && aload(&ctl->abort) == 0u) {
        if (atomicAdd(&ctl->leave, 1u) + 1u == gridDim.x) {
            for (unsigned p = 0; p < n_pairs; p++) {
                NdtPoolPair *P = pair_at(p);
                __hip_atomic_store(&P->ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&P->done_tasks, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __hip_atomic_store(&ctl->finished, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->leave, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// One launch for the whole batch; `work` = a control block + per registration a header and one row per chunk the source
// set can hold (ndt_match_pool_pair_bytes); headers and control block zero.
hipError_t ndt_launch_match_pool(const NdtSetView &tset, const uint32_t *tidx_dev, const NdtSetView &sset,
                                 const uint32_t *sidx_dev, double *T16_dev, size_t n_pairs, const NdtMatchParamsDev &prm,
                                 NdtMatchResultDev *res_dev, const double *Q36_dev, unsigned n_groups,
                                 unsigned cells_per_group, void *work_dev, size_t pair_stride, hipStream_t stream)
{
    char *work = (char *)work_dev;
    const dim3 grid(n_groups), block(NDT_MATCH_THREADS);
#define NDT_LAUNCH_POOL(NN)                                                                                            \
    hipLaunchKernelGGL(ndt_match_pool_kernel<NN>, grid, block, 0, stream, tset, tidx_dev, sset, sidx_dev, T16_dev, prm,  \
                       res_dev, Q36_dev, work, pair_stride, cells_per_group, (unsigned)n_pairs)
    switch (prm.n_neighbours) {
    case 0: NDT_LAUNCH_POOL(0); break;
    case 1: NDT_LAUNCH_POOL(1); break;
    case 2: NDT_LAUNCH_POOL(2); break;
    case 3: NDT_LAUNCH_POOL(3); break;
    default: return hipErrorInvalidValue;
    }
#undef NDT_LAUNCH_POOL
    return hipGetLastError();
}

hipError_t ndt_launch_eval(const NdtSetView &tset, size_t tmap, const NdtSetView &sset, size_t smap, const rigid &T,
                           int n_neighbours, int with_h, double lfd1, double lfd2, unsigned n_groups, double *partials_dev,
                           hipStream_t stream)
{
#define NDT_LAUNCH_EVAL(NN)                                                                                          \
    hipLaunchKernelGGL(ndt_eval_kernel<NN>, dim3(n_groups), dim3(NDT_MATCH_THREADS), 0, stream, tset, (unsigned)tmap, \
                       sset, (unsigned)smap, T, with_h, lfd1, lfd2, partials_dev)
    switch (n_neighbours) {
    case 0: NDT_LAUNCH_EVAL(0); break;
    case 1: NDT_LAUNCH_EVAL(1); break;
    case 2: NDT_LAUNCH_EVAL(2); break;
    case 3: NDT_LAUNCH_EVAL(3); break;
    default: return hipErrorInvalidValue;
    }
#undef NDT_LAUNCH_EVAL
    return hipGetLastError();
}

hipError_t ndt_launch_match(const NdtSetView &tset, const uint32_t *tidx_dev, const NdtSetView &sset,
                            const uint32_t *sidx_dev, double *T16_dev, size_t n_pairs, const NdtMatchParamsDev &prm,
                            NdtMatchResultDev *res_dev, const double *Q36_dev, const unsigned *feat_off_dev,
                            const double *feat_cells_dev, unsigned n_groups, int park_iters,
                            int slots, unsigned double_thresh, void *work_dev, hipStream_t stream)
{
    if (n_pairs == 0) return hipSuccess;
    if (slots < 1 || slots > 3) return hipErrorInvalidValue;
    // ticket counters and the parked list start at zero
    hipError_t e = hipMemsetAsync(work_dev, 0, sizeof(NdtMatchWork) + (n_pairs + (size_t)n_groups * slots + 1) * sizeof(unsigned), stream);
    if (e != hipSuccess) return e;
#define NDT_LAUNCH_MATCH(NN)                                                                                         \
    do {                                                                                                             \
        if (slots == 3)                                                                                              \
            hipLaunchKernelGGL((ndt_match_kernel<NN, 3>), dim3(n_groups), dim3(NDT_MATCH_THREADS), 0, stream, tset, tidx_dev, sset, \
                               sidx_dev, T16_dev, prm, res_dev, Q36_dev, feat_off_dev, feat_cells_dev, (unsigned)n_pairs, park_iters, double_thresh, (char *)work_dev); \
        else if (slots == 2)                                                                                         \
            hipLaunchKernelGGL((ndt_match_kernel<NN, 2>), dim3(n_groups), dim3(NDT_MATCH_THREADS), 0, stream, tset, tidx_dev, sset, \
                               sidx_dev, T16_dev, prm, res_dev, Q36_dev, feat_off_dev, feat_cells_dev, (unsigned)n_pairs, park_iters, double_thresh, (char *)work_dev); \
        else                                                                                                         \
            hipLaunchKernelGGL((ndt_match_kernel<NN, 1>), dim3(n_groups), dim3(NDT_MATCH_THREADS), 0, stream, tset, tidx_dev, sset, \
                               sidx_dev, T16_dev, prm, res_dev, Q36_dev, feat_off_dev, feat_cells_dev, (unsigned)n_pairs, park_iters, double_thresh, (char *)work_dev); \
    } while (0)
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

#ifdef NDT_MATCH_TL
extern "C" int ndtgpu_debug_timeline(long long *out, int reset)      // out: 4 * 4096 + 1024 + 8 values
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    const size_t n = 4 * NDT_TL_PAIRS + 1024 + 8;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl), n * sizeof(long long)) != hipSuccess) return -1;
    if (reset) {
        static long long z[4 * NDT_TL_PAIRS + 1024 + 8];
        for (size_t k = 0; k < n; k++) z[k] = 0;
        z[4 * NDT_TL_PAIRS + 1024] = 0x7fffffffffffffffll;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_tl), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#ifdef NDT_MATCH_PROF
// n evaluations of one pair at one pose in one workgroup (wide form): shader clocks per evaluation
template <int NN>
__global__ __launch_bounds__(NDT_MATCH_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void ndt_eval_loop_kernel(
    NdtSetView tset, unsigned tmap, NdtSetView sset, unsigned smap, rigid T, int with_h, int n_iter, int reuse, double lfd1,
    double lfd2, long long *out)
{
    __shared__ EvalShared<NDT_MATCH_WAVES> sh;
    __shared__ rigid s_T;
    const MapView tg = map_view(tset, tmap);
    const MapView sv = map_view(sset, smap);
    if (threadIdx.x == 0) s_T = T;
    if (threadIdx.x < NDT_VW) sh.cache[threadIdx.x].key = 0u;
    __syncthreads();
    const long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < n_iter; it++) {
        if (with_h) eval_derivs<NN, true>(tg, sv.cells, sv.n_cells, s_T, lfd1, lfd2, sh, reuse ? 7u : 0u);
        else eval_derivs<NN, false>(tg, sv.cells, sv.n_cells, s_T, lfd1, lfd2, sh, reuse ? 7u : 0u);
    }
    const long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = c1 - c0; out[blockIdx.x * 4 + 1] = (long long)sh.sums[28]; out[blockIdx.x * 4 + 2] = sv.n_cells; }
}
extern "C" int ndtgpu_debug_eval_loop(void *tsv, unsigned tmap, void *ssv, unsigned smap, const double *T16, int with_h, int n_iter,
                                      int reuse, int n_groups, double lfd1, double lfd2, long long *out_host)
{
    const NdtSetView &tset = *reinterpret_cast<const NdtSetView *>(tsv), &sset = *reinterpret_cast<const NdtSetView *>(ssv);
    rigid T;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T.r[r * 3 + c] = T16[c * 4 + r]; T.t[r] = T16[12 + r]; }
    long long *out_dev = nullptr;
    if (hipMalloc(&out_dev, n_groups * 4 * sizeof(long long)) != hipSuccess) return -1;
    hipLaunchKernelGGL(ndt_eval_loop_kernel<2>, dim3(n_groups), dim3(NDT_MATCH_THREADS), 0, 0, tset, tmap, sset, smap, T, with_h, n_iter,
                       reuse, lfd1, lfd2, out_dev);
    int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -2;
    if (rc == 0 && hipMemcpy(out_host, out_dev, n_groups * 4 * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) rc = -3;
    hipFree(out_dev);
    return rc;
}

extern "C" int ndtgpu_debug_solver_prof(long long out[16], int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_solver_prof), 16 * sizeof(long long)) != hipSuccess) return -1;
    if (reset) {
        long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_solver_prof), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
extern "C" int ndtgpu_debug_wave_prof(long long out[16], int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave), 16 * sizeof(long long)) != hipSuccess) return -1;
    if (reset) {
        long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_wave), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
extern "C" int ndtgpu_debug_prof(long long out[16], int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), 16 * sizeof(long long)) != hipSuccess) return -1;
    if (reset) {
        long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

hipError_t ndt_launch_covariance(const NdtSetView &tset, const uint32_t *tidx_dev, const NdtSetView &sset,
                                 const uint32_t *sidx_dev, const double *T16_dev, size_t n_links, int n_neighbours,
                                 double lfd1, double lfd2, int mode, double *cov36_dev, int *status_dev, hipStream_t stream)
{
    if (n_links == 0) return hipSuccess;
#define NDT_LAUNCH_COV(NN)                                                                                            \
    hipLaunchKernelGGL(ndt_covariance_kernel<NN>, dim3((unsigned)n_links), dim3(NDT_MATCH_THREADS), 0, stream, tset,  \
                       tidx_dev, sset, sidx_dev, T16_dev, lfd1, lfd2, mode, cov36_dev, status_dev)
    switch (n_neighbours) {
    case 0: NDT_LAUNCH_COV(0); break;
    case 1: NDT_LAUNCH_COV(1); break;
    case 2: NDT_LAUNCH_COV(2); break;
    case 3: NDT_LAUNCH_COV(3); break;
    default: return hipErrorInvalidValue;
    }
#undef NDT_LAUNCH_COV
    return hipGetLastError();
}
