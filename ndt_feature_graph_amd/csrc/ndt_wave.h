// ndt_wave.h -- cross-lane primitives of a 64-wide wavefront on gfx950 (product code, device only): everything here runs
// in the vector ALU (DPP modifiers, v_permlane32_swap / v_permlane16_swap), no LDS round trips.
#pragma once
#include "ndt_common.h"

// ballot of a lane predicate as the compare's own lane mask (HIP's __ballot(int) first turns the predicate into 0 / 1 in a
// vector register and compares that with zero: two vector instructions per ballot)
NDT_D unsigned long long ndt_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// inclusive scan over the 64 lanes (DPP row shifts, then the two row broadcasts)
NDT_D unsigned ndt_wave_incl_scan(unsigned v)
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

// a + b where, afterwards, the lanes whose bit 5 (rows16 = false) or bit 4 (rows16 = true) is clear hold the sum of a
// over the lane pair (l, l ^ 32 / 16) and the other lanes the sum of b: one swap per 32-bit half and ONE add for two
// values
NDT_D double pl_swap_add(double a, double b, bool rows16)
{
    const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    if (rows16) {
        auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    }
    auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}

// value of lane (l ^ O) for O = 1, 2, 4, 8 in the vector ALU: quad permutes for 1 and 2; lane ^ 4 is the half-row mirror
// (^ 7) of the quad reversal (^ 3), lane ^ 8 the row mirror (^ 15) of the half-row mirror (^ 7).  No LDS round trip.
template <int O>
NDT_D double xor_lane(double x)
{
    static_assert(O == 1 || O == 2 || O == 4 || O == 8, "within a row of 16 lanes");
    int lo = __double2loint(x), hi = __double2hiint(x);
    if constexpr (O == 1) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
        hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, true);
    } else if constexpr (O == 2) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, true);
    } else if constexpr (O == 4) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x1B, 0xf, 0xf, true);    // quad_perm [3,2,1,0]
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x1B, 0xf, 0xf, true);
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xf, 0xf, true);   // row_half_mirror
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xf, 0xf, true);
    } else {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xf, 0xf, true);   // row_half_mirror
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xf, 0xf, true);
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xf, 0xf, true);   // row_mirror
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xf, 0xf, true);
    }
    return __hiloint2double(hi, lo);
}

// Sum of N (power of two) per-lane values over the 64 lanes of a wave, all N at once: a butterfly that halves the
// number of values a lane carries at every step (the lane keeps the half selected by its lane bit and receives
// the partner's copy of it).  2N - 1 exchanges instead of 6N, and the association order is the one of the plain xor tree
// v += shfl_xor(v, 32), 16, ..., 1 -- the result is bit-identical to it.
template <int N, int HALF, int O>
NDT_D void wave_sum_step(double (&v)[N], unsigned lane)
{
#pragma unroll
    for (int k = 0; k < HALF; k++) {
        if constexpr (O >= 16) {
            v[k] = pl_swap_add(v[k], v[k + HALF], O == 16);
        } else {
            const bool up = (lane & (unsigned)O) != 0;
            const double keep = up ? v[k + HALF] : v[k], send = up ? v[k] : v[k + HALF];
            v[k] = keep + xor_lane<O>(send);
        }
    }
    if constexpr (HALF > 1) wave_sum_step<N, HALF / 2, O / 2>(v, lane);   // static indices only: v stays in registers
}

// The nine moments {sum d (3), sum d d^T (6)} of a run, summed over the wave: 9 -> 5 -> 3 values by the two lane swaps,
// 3 -> 2 -> 1 by selects + DPP, then the two low lane bits.  Lane l ends with the total of moment
// ndt_moment_of_lane(l) (or of a zero pad): 57 instructions for the nine sums.
NDT_D double wave_sum_moments(const double (&sd)[3], const double (&se)[6], unsigned lane)
{
    const double v0 = sd[0], v1 = sd[1], v2 = sd[2], v3 = se[0], v4 = se[1], v5 = se[2], v6 = se[3], v7 = se[4], v8 = se[5];
    const double zero = 0.0;
    // bit 5: (v0 | v5) (v1 | v6) (v2 | v7) (v3 | v8) (v4 | 0)
    const double w0 = pl_swap_add(v0, v5, false), w1 = pl_swap_add(v1, v6, false), w2 = pl_swap_add(v2, v7, false),
                 w3 = pl_swap_add(v3, v8, false), w4 = pl_swap_add(v4, zero, false);
    // bit 4: (w0 | w3) (w1 | w4) (w2 | 0)
    const double x0 = pl_swap_add(w0, w3, true), x1 = pl_swap_add(w1, w4, true), x2 = pl_swap_add(w2, zero, true);
    // bit 3: (x0 | x1), x2 alone
    const bool b3 = (lane & 8u) != 0, b2 = (lane & 4u) != 0;
    const double y0 = (b3 ? x1 : x0) + xor_lane<8>(b3 ? x0 : x1);
    const double y1 = x2 + xor_lane<8>(x2);
    // bit 2: (y0 | y1)
    double z = (b2 ? y1 : y0) + xor_lane<4>(b2 ? y0 : y1);
    z += xor_lane<2>(z);
    z += xor_lane<1>(z);
    return z;
}

// which moment (0..2 = sum d, 3..8 = sum d d^T) wave_sum_moments leaves in lane l, for the ONE lane per moment that
// hands it on (-1 for every other lane: copies, zero pads)
NDT_HD int ndt_moment_of_lane(unsigned l)
{
    const unsigned b5 = (l >> 5) & 1u, b4 = (l >> 4) & 1u, b3 = (l >> 3) & 1u, b2 = (l >> 2) & 1u;
    if (l & 3u) return -1;          // the two low lane bits hold copies
    if (b2) {                       // x2 lineage: (w2 | 0) over bit 4, w2 = (v2 | v7) over bit 5; both b3 hold it
        if (b4 || b3) return -1;
        return b5 ? 7 : 2;
    }
    if (!b3) {                      // x0 lineage: (w0 | w3), w0 = (v0 | v5), w3 = (v3 | v8)
        return b4 ? (b5 ? 8 : 3) : (b5 ? 5 : 0);
    }
    // x1 lineage: (w1 | w4), w1 = (v1 | v6), w4 = (v4 | 0)
    return b4 ? (b5 ? -1 : 4) : (b5 ? 6 : 1);
}
