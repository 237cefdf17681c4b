// ndt_binning.h -- point -> LazyGrid cell for the build kernels (product code, device only).
//
// LazyGrid::getIndexForPoint (perception_oru; call sites ndt_feature_fuser_hmt.cpp:195-226) evaluates, per axis,
// floor((p - c) / res + 0.5) + size / 2.0 in fp64 and truncates to int; NDTMap::loadPointCloud drops NaN points and, with
// a range limit, points with |p - origin| > range.  fast() evaluates both in fp32 with derived error bounds and reports the
// points whose fp32 result could differ from the reference's; exact() decides those with the reference's own formulas.
// Binning is therefore bit-identical to the reference's for every point.
#pragma once
#include "ndt_math.h"

struct NdtBinner {
    double cx, cy, cz, res, ox, oy, oz, range_limit;
    double hx, hy, hz;          // size / 2.0 (NdtGrid::half: scalar registers)
    int sx, sy, sz;
    float inv32, kx32, ky32, kz32, ox32, oy32, oz32, r2, r2eff, r2band, frac_lim, z_max32;

    // centre: the map's grid centre, origin: the range test's origin (the sensor; 0 without one)
    NDT_D void init(const NdtGrid &g, double cx_, double cy_, double cz_, double ox_, double oy_, double oz_,
                    double range_limit_, float z_max32_)
    {
        cx = cx_; cy = cy_; cz = cz_; res = g.res; ox = ox_; oy = oy_; oz = oz_; range_limit = range_limit_;
        sx = g.size[0]; sy = g.size[1]; sz = g.size[2];
        z_max32 = z_max32_;
        const double inv_res = 1.0 / g.res;
        hx = g.half[0]; hy = g.half[1]; hz = g.half[2];
        inv32 = (float)inv_res;
        // fast-path constants: idx = floor(p*inv + k), k = 0.5 + size/2 - c*inv.  Only for EVEN sizes (size/2
        // integral); with an odd size the reference's double->int truncation makes the index formula
        // non-monotone, so every point takes the exact path then (force_exact).
        const bool force_exact = ((g.size[0] | g.size[1] | g.size[2]) & 1) != 0;
        kx32 = (float)(0.5 + hx - cx * inv_res); ky32 = (float)(0.5 + hy - cy * inv_res); kz32 = (float)(0.5 + hz - cz * inv_res);
        ox32 = (float)ox; oy32 = (float)oy; oz32 = (float)oz;
        r2 = (float)(range_limit * range_limit);
        r2eff = range_limit > 0 ? r2 : __builtin_inff();
        r2band = range_limit > 0 ? 1e-3f * r2 : -1.0f;
        // fp32 error of v = fma(p, inv32, k32) against (p - c)/res + 0.5 + size/2 for a point in or next to the grid:
        // inv32, k32 and the fma each round once (2^-24 relative), |p/res| <= |v| + |k|  =>  |error| <= 1.2e-7 (|v| + |k|),
        // |v| <= size + 1.  Points whose fraction is within twice that bound of a cell face take the exact path; a
        // point further outside the grid is out of bounds on either path (error < 1 cell up to 2^23 cells, above
        // that the float -> int conversion saturates).
        const float kmax = fmaxf(fmaxf(fabsf(kx32), fabsf(ky32)), fabsf(kz32));
        const float smax = (float)max(max(g.size[0], g.size[1]), g.size[2]) + 1.0f;
        const float face_guard = 2.4e-7f * (smax + kmax);
        // fast path: |frac - 0.5| <= frac_lim on every axis; odd sizes / absurd centres: exact path for every point
        frac_lim = (force_exact || !(face_guard < 0.25f)) ? -1.0f : 0.5f - face_guard;
    }

    // The fp32 constants come out of the vector ALU and would each occupy a vector register although they are the same
    // in every lane; a readfirstlane moves them to scalar registers (for kernels whose vector registers are scarce).
    NDT_D void scalarize()
    {
        float *f[] = {&inv32, &kx32, &ky32, &kz32, &ox32, &oy32, &oz32, &r2, &r2eff, &r2band, &frac_lim, &z_max32};
#pragma unroll
        for (int k = 0; k < 12; k++) *f[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(*f[k])));
    }

    // fp32 fast path of one point -> cell index.  Every test is false for a NaN (NaN points are skipped, like padding);
    // an Inf passes the range test when no range is set and is then out of the grid.  Returns true when the
    // reference's fp64 formulas must decide (exact()): the point is within the fp32 error bound of a cell face or
    // of the range sphere (frac_lim < 0 sends every point there: odd grid sizes, absurd centres).
    // (the cell index leaves as the three floats floor() made: the offset arithmetic wants them as floats again)
    NDT_D bool fast(float fx, float fy, float fz, float &gx, float &gy, float &gz, int &slot) const
    {
        const float dx = fx - ox32, dy = fy - oy32, dz = fz - oz32;
        const float dd = dx * dx + dy * dy + dz * dz;
        const bool okz = fz <= z_max32;                                 // addPointCloud's maxz (+inf otherwise)
        const bool okr = (dd <= r2eff) & okz;                           // r2eff = +inf without a range limit
        const bool near_r = fabsf(dd - r2) < r2band;                    // r2band < 0 without a range limit
        // v = (p - c)/res + 0.5 + size/2 in one fma per axis; the integer part is the cell index
        const float vx = fmaf(fx, inv32, kx32), vy = fmaf(fy, inv32, ky32), vz = fmaf(fz, inv32, kz32);
        const float flx = floorf(vx), fly = floorf(vy), flz = floorf(vz);
        const float tx = fabsf((vx - flx) - 0.5f), ty = fabsf((vy - fly) - 0.5f), tz = fabsf((vz - flz) - 0.5f);
        const int ix = (int)flx, iy = (int)fly, iz = (int)flz;
        gx = flx; gy = fly; gz = flz;
        const bool inb = ((unsigned)ix < (unsigned)sx) & ((unsigned)iy < (unsigned)sy) & ((unsigned)iz < (unsigned)sz);
        const int sl = (int)(((unsigned)ix * (unsigned)sy + (unsigned)iy) * (unsigned)sz + (unsigned)iz);
        slot = (okr & inb) ? sl : -1;
        return (okr | (near_r & okz)) & (near_r | !(fmaxf(fmaxf(tx, ty), tz) <= frac_lim));
    }

    // ... and the reference's own formulas for those few points
    NDT_D void exact(float fx, float fy, float fz, float &gx, float &gy, float &gz, int &slot) const
    {
        int ix = (int)gx, iy = (int)gy, iz = (int)gz;
        const float dx = fx - ox32, dy = fy - oy32, dz = fz - oz32;
        const float dd = dx * dx + dy * dy + dz * dz;
        const bool okz = fz <= z_max32;
        bool ok = (dd <= r2eff) & okz;
        if (fabsf(dd - r2) < r2band) {
#pragma clang fp contract(off)
            double ex = (double)fx - ox, ey = (double)fy - oy, ez = (double)fz - oz;
            ok = !(sqrt(ex * ex + ey * ey + ez * ez) > range_limit) && okz;
        }
        const float vx = fmaf(fx, inv32, kx32), vy = fmaf(fy, inv32, ky32), vz = fmaf(fz, inv32, kz32);
        if (!(fabsf((vx - floorf(vx)) - 0.5f) <= frac_lim)) ix = lazygrid_index_half((double)fx, cx, res, hx);
        if (!(fabsf((vy - floorf(vy)) - 0.5f) <= frac_lim)) iy = lazygrid_index_half((double)fy, cy, res, hy);
        if (!(fabsf((vz - floorf(vz)) - 0.5f) <= frac_lim)) iz = lazygrid_index_half((double)fz, cz, res, hz);
        slot = (ok && (unsigned)ix < (unsigned)sx && (unsigned)iy < (unsigned)sy && (unsigned)iz < (unsigned)sz)
                   ? (int)(((unsigned)ix * (unsigned)sy + (unsigned)iy) * (unsigned)sz + (unsigned)iz) : -1;
        gx = (float)ix; gy = (float)iy; gz = (float)iz;           // (only cells of the grid matter: exact below 2^24)
    }
};

// rint(t) of a double |t| < 2^62 as a 64-bit integer (there is no fp64 -> int64 conversion instruction): the upper word
// by floor(t 2^-32), the lower one from the exact remainder in [0, 2^32)
NDT_D long long ndt_fixed_from_double(double t)
{
    double hi = floor(t * (1.0 / 4294967296.0));
    double lo = rint(fma(-hi, 4294967296.0, t));            // exact remainder, rounded to an integer in [0, 2^32]
    const bool carry = lo >= 4294967296.0;
    lo = carry ? 0.0 : lo;
    const long long h = (long long)(int)hi + (carry ? 1 : 0);
    return (long long)((unsigned long long)h << 32) + (long long)(unsigned long long)(unsigned)lo;
}

// NDTCell::computeGaussian (first Gaussian of a cell) + rescaleCovariance on the exact moments of a cell: n points,
// s1 = sum u * 2^s1_shift, s2 = sum u u^T * 2^s2_shift with u = (p - cell centre) / res.  `centre` = the cell's centre
// (c + (k - size/2) res per axis).  Returns a record with n == 0 when the cell gets no Gaussian (fewer than n_min
// points, or a rank-deficient covariance: NDT_DEGENERATE_REL).
NDT_D NdtCell ndt_gaussian_from_moments(const NdtAcc &a, unsigned slot, const double centre[3], double res, int n_min,
                                        double eval_factor, double IS1, double IS2)
{
    NdtCell c;
    c.n = 0;
    c.slot = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) c.mean[k] = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) c.cov[k] = 0;
    const unsigned long long n = (unsigned long long)a.n;
    if (n >= 2 && n >= (unsigned long long)n_min) {
        double dn = (double)n;
        double m[3];   // mean offset in cell units
#pragma unroll
        for (int k = 0; k < 3; k++) m[k] = ((double)a.s1[k] / dn) * IS1;
        double S[6];
#pragma unroll
        for (int k = 0; k < 6; k++) S[k] = (double)a.s2[k] * IS2;
        double sc = res * res / (dn - 1.0);
        double C[3][3], V[3][3];
        C[0][0] = (S[0] - dn * m[0] * m[0]) * sc;
        C[0][1] = (S[1] - dn * m[0] * m[1]) * sc;
        C[0][2] = (S[2] - dn * m[0] * m[2]) * sc;
        C[1][1] = (S[3] - dn * m[1] * m[1]) * sc;
        C[1][2] = (S[4] - dn * m[1] * m[2]) * sc;
        C[2][2] = (S[5] - dn * m[2] * m[2]) * sc;
        C[1][0] = C[0][1]; C[2][0] = C[0][2]; C[2][1] = C[1][2];
        double E[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int q2 = 0; q2 < 3; q2++) E[r][q2] = C[r][q2];
        jacobi_static<3, true>(E, V);
        double ev[3] = {E[0][0], E[1][1], E[2][2]};
        double mx = dmax3(ev[0], ev[1], ev[2]), mn = dmin3(ev[0], ev[1], ev[2]);
        // NDTCell::rescaleCovariance
        if (mx > 0 && mn > NDT_DEGENERATE_REL * mx) {
            bool recalc = false;
#pragma unroll
            for (int k = 0; k < 3; k++)
                if (mx > ev[k] * eval_factor) { ev[k] = mx / eval_factor; recalc = true; }
            if (recalc) {
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int q2 = r; q2 < 3; q2++) {
                        double s = 0;
#pragma unroll
                        for (int k = 0; k < 3; k++) s += V[r][k] * ev[k] * V[q2][k];
                        C[r][q2] = s;
                    }
            }
            c.mean[0] = centre[0] + m[0] * res;
            c.mean[1] = centre[1] + m[1] * res;
            c.mean[2] = centre[2] + m[2] * res;
            c.cov[0] = C[0][0]; c.cov[1] = C[0][1]; c.cov[2] = C[0][2];
            c.cov[3] = C[1][1]; c.cov[4] = C[1][2]; c.cov[5] = C[2][2];
            c.n = (uint32_t)n;
            c.slot = slot;
        }
    }
    return c;
}
