// ndt_math.h -- fixed-size fp64 algebra for the NDT kernels (host + device, no Eigen).
#pragma once
#include "ndt_common.h"
#include <math.h>
#include <utility>

struct d3 { double x, y, z; };

NDT_HD d3 operator+(d3 a, d3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
NDT_HD d3 operator-(d3 a, d3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
NDT_HD d3 operator*(double s, d3 a) { return {s * a.x, s * a.y, s * a.z}; }
NDT_HD double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
NDT_HD d3 cross(d3 a, d3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// e_k x v
NDT_HD d3 ex_cross(d3 v) { return {0.0, -v.z, v.y}; }
NDT_HD d3 ey_cross(d3 v) { return {v.z, 0.0, -v.x}; }
NDT_HD d3 ez_cross(d3 v) { return {-v.y, v.x, 0.0}; }

NDT_HD double dmax3(double a, double b, double c) { double m = a > b ? a : b; return m > c ? m : c; }
NDT_HD double dmin3(double a, double b, double c) { double m = a < b ? a : b; return m < c ? m : c; }

struct sym3 { double xx, xy, xz, yy, yz, zz; };
NDT_HD sym3 operator+(sym3 a, sym3 b) { return {a.xx + b.xx, a.xy + b.xy, a.xz + b.xz, a.yy + b.yy, a.yz + b.yz, a.zz + b.zz}; }
NDT_HD d3 mul(sym3 a, d3 v)
{
    return {a.xx * v.x + a.xy * v.y + a.xz * v.z, a.xy * v.x + a.yy * v.y + a.yz * v.z,
            a.xz * v.x + a.yz * v.y + a.zz * v.z};
}

// (C_i + C_j)^-1 with Eigen's computeInverseAndDetWithCheck semantics (|det| > 1e-12)
NDT_HD bool inverse_check(sym3 a, sym3 &inv)
{
    double c00 = a.yy * a.zz - a.yz * a.yz;
    double c01 = a.yz * a.xz - a.xy * a.zz;
    double c02 = a.xy * a.yz - a.yy * a.xz;
    double det = a.xx * c00 + a.xy * c01 + a.xz * c02;
    if (!(fabs(det) > NDT_DET_EPS)) return false;
    double id = 1.0 / det;
    inv.xx = c00 * id;
    inv.xy = c01 * id;
    inv.xz = c02 * id;
    inv.yy = (a.xx * a.zz - a.xz * a.xz) * id;
    inv.yz = (a.xy * a.xz - a.xx * a.yz) * id;
    inv.zz = (a.xx * a.yy - a.xy * a.xy) * id;
    return true;
}

struct rigid { double r[9]; double t[3]; };   // row-major rotation + translation
static_assert(sizeof(rigid) == 12 * sizeof(double), "rigid is twelve consecutive doubles (compose_pose_wave writes entry e of them)");

NDT_HD d3 apply(const rigid &T, d3 p)
{
    return {T.r[0] * p.x + T.r[1] * p.y + T.r[2] * p.z + T.t[0], T.r[3] * p.x + T.r[4] * p.y + T.r[5] * p.z + T.t[1],
            T.r[6] * p.x + T.r[7] * p.y + T.r[8] * p.z + T.t[2]};
}

// R C R^T for symmetric C
NDT_HD sym3 rotate_cov(const double *R, sym3 c)
{
    double a[9]; // a = R C
    for (int i = 0; i < 3; i++) {
        a[i * 3 + 0] = R[i * 3] * c.xx + R[i * 3 + 1] * c.xy + R[i * 3 + 2] * c.xz;
        a[i * 3 + 1] = R[i * 3] * c.xy + R[i * 3 + 1] * c.yy + R[i * 3 + 2] * c.yz;
        a[i * 3 + 2] = R[i * 3] * c.xz + R[i * 3 + 1] * c.yz + R[i * 3 + 2] * c.zz;
    }
    sym3 o;
    o.xx = a[0] * R[0] + a[1] * R[1] + a[2] * R[2];
    o.xy = a[0] * R[3] + a[1] * R[4] + a[2] * R[5];
    o.xz = a[0] * R[6] + a[1] * R[7] + a[2] * R[8];
    o.yy = a[3] * R[3] + a[4] * R[4] + a[5] * R[5];
    o.yz = a[3] * R[6] + a[4] * R[7] + a[5] * R[8];
    o.zz = a[6] * R[6] + a[7] * R[7] + a[8] * R[8];
    return o;
}

// sin and cos of a pose-increment angle.  Newton / line-search increments are small: for |x| <= pi/4 the minimax
// kernels of fdlibm (k_sin.c / k_cos.c in the msun form, error < 1 ulp) need no argument reduction -- ~25
// instructions against ~300 for the general library pair, and the serial solver composes three of them per trial
// pose.  Larger angles take the library functions.
NDT_HD void sincos_pose(double x, double &sn, double &cs)
{
    // (one fixed sequence of operations wherever this is inlined: explicit fma, no other contraction)
#pragma clang fp contract(off)
    if (fabs(x) <= 0.78539816339744830962) {
        const double z = x * x, w = z * z;
        const double rs = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                     -1.98412698298579493134e-04), 8.33333333332248946124e-03);
        sn = fma(z * x, fma(z, rs, -1.66666666666666324348e-01), x);
        const double rc = fma(w * w, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                              z * fma(z, fma(z, 2.48015872894767294178e-05, -1.38888888888741095749e-03), 4.16666666666666019037e-02));
        const double hz = 0.5 * z, c1 = 1.0 - hz;
        cs = c1 + fma(z, rc, (1.0 - c1) - hz);
    } else {
        sn = sin(x);
        cs = cos(x);
    }
}

// TR = Translation(p0,p1,p2) * Rx(p3) * Ry(p4) * Rz(p5)   (ndt_matcher_d2d_fusion.h:1036-1039)
// Every entry is written as ONE fixed sequence of operations (explicit fma, no expression the compiler may contract one way
// here and another way there): the persistent matchers compute the twelve entries on twelve lanes (compose_pose_wave in
// ndt_match.hip, from the same operands through the same operations), everybody else on one, and the bits agree.
NDT_HD void pose_to_rigid(const double *p, rigid &T)
{
    double cx, sx, cy, sy, cz, sz;
    sincos_pose(p[3], sx, cx);
    sincos_pose(p[4], sy, cy);
    sincos_pose(p[5], sz, cz);
    const double P = sx * sy, Q = cx * sy;
    // Rx*Ry*Rz
    T.r[0] = cy * cz;               T.r[1] = (-cy) * sz;             T.r[2] = sy;
    T.r[3] = fma(P, cz, cx * sz);   T.r[4] = fma(-P, sz, cx * cz);   T.r[5] = (-sx) * cy;
    T.r[6] = fma(-Q, cz, sx * sz);  T.r[7] = fma(Q, sz, sx * cz);    T.r[8] = cx * cy;
    T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2];
}

// C = A * B (apply B first)
NDT_HD void rigid_mul(const rigid &A, const rigid &B, rigid &C)
{
    rigid o;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            o.r[i * 3 + j] = fma(A.r[i * 3 + 2], B.r[6 + j], fma(A.r[i * 3 + 1], B.r[3 + j], A.r[i * 3] * B.r[j]));
        o.t[i] = fma(A.r[i * 3 + 2], B.t[2], fma(A.r[i * 3 + 1], B.t[1], A.r[i * 3] * B.t[0])) + A.t[i];
    }
    C = o;
}

// cyclic Jacobi eigensolver, symmetric n x n (n <= 6), row-major; evals ascending, evecs in columns.
// Stands in for Eigen::SelfAdjointEigenSolver (fusion.h:922-928; NDTCell::rescaleCovariance).
template <int NMAX>
NDT_HD void jacobi_eig(int n, const double *A, double *evals, double *V)
{
    double a[NMAX * NMAX];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            a[i * NMAX + j] = 0.5 * (A[i * n + j] + A[j * n + i]);
            V[i * n + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; i++) {
            diag += a[i * NMAX + i] * a[i * NMAX + i];
            for (int j = i + 1; j < n; j++) off += a[i * NMAX + j] * a[i * NMAX + j];
        }
        if (off == 0.0 || off <= 1e-60 * diag) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = a[p * NMAX + q];
                if (apq == 0.0) continue;
                double theta = (a[q * NMAX + q] - a[p * NMAX + p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = a[k * NMAX + p], akq = a[k * NMAX + q];
                    a[k * NMAX + p] = c * akp - s * akq;
                    a[k * NMAX + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = a[p * NMAX + k], aqk = a[q * NMAX + k];
                    a[p * NMAX + k] = c * apk - s * aqk;
                    a[q * NMAX + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    // selection sort of (eval, column)
    for (int i = 0; i < n; i++) evals[i] = a[i * NMAX + i];
    for (int i = 0; i < n; i++) {
        int m = i;
        for (int j = i + 1; j < n; j++)
            if (evals[j] < evals[m]) m = j;
        if (m != i) {
            double t = evals[i]; evals[i] = evals[m]; evals[m] = t;
            for (int k = 0; k < n; k++) { double u = V[k * n + i]; V[k * n + i] = V[k * n + m]; V[k * n + m] = u; }
        }
    }
}

// Same cyclic Jacobi with compile-time size and fully unrolled loops: every index is a constant, so
// the matrices live in registers (no scratch).  On exit the diagonal of `a` holds the eigenvalues
// (unsorted) and, when WANT_V, the columns of `v` the eigenvectors.
template <int N, bool WANT_V>
NDT_HD void jacobi_static(double (&a)[N][N], double (&v)[N][N])
{
    if (WANT_V) {
#pragma unroll
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int j = 0; j < N; j++) v[i][j] = (i == j) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = i + 1; j < N; j++) {
            double s = 0.5 * (a[i][j] + a[j][i]);
            a[i][j] = s;
            a[j][i] = s;
        }
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0, diag = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            diag += a[i][i] * a[i][i];
#pragma unroll
            for (int j = i + 1; j < N; j++) off += a[i][j] * a[i][j];
        }
        // off-diagonal mass below 1e-15 of the diagonal: eigenvalue errors ~ off^2/gap < 1e-30 relative
        if (off == 0.0 || off <= 1e-30 * diag) break;
#pragma unroll
        for (int p = 0; p < N; p++) {
#pragma unroll
            for (int q = p + 1; q < N; q++) {
                double apq = a[p][q];
                if (apq != 0.0) {
                    double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                    double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        double akp = a[k][p], akq = a[k][q];
                        a[k][p] = c * akp - s * akq;
                        a[k][q] = s * akp + c * akq;
                    }
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        double apk = a[p][k], aqk = a[q][k];
                        a[p][k] = c * apk - s * aqk;
                        a[q][k] = s * apk + c * aqk;
                    }
                    if (WANT_V) {
#pragma unroll
                        for (int k = 0; k < N; k++) {
                            double vkp = v[k][p], vkq = v[k][q];
                            v[k][p] = c * vkp - s * vkq;
                            v[k][q] = s * vkp + c * vkq;
                        }
                    }
                }
            }
        }
    }
}

// lambda_min and lambda_max of a symmetric 6x6 matrix (what the Newton loop's regulariser needs from
// Eigen::SelfAdjointEigenSolver, fusion.h:922-940) without a full eigen-decomposition: Householder
// tridiagonalisation (4 reflectors, static indices: registers only) and Laguerre's iteration on the
// characteristic polynomial of the tridiagonal matrix, started outside the spectrum at the Gershgorin bounds.
// For a polynomial with only real roots Laguerre converges monotonically (and cubically) to the nearest root from
// outside, i.e. to lambda_min from the left and lambda_max from the right.  The three-term recurrence is the
// Sturm sequence (backward stable): the results carry an error of a few eps * ||A||, like the cyclic Jacobi
// sweeps this replaces (~60 k shader cycles on one lane against ~6 k).
namespace ndt_eig6 {
// p, p', p'' of det(T - x I) for the tridiagonal T = (d, e) by the three-term recurrence
NDT_HD void charpoly(const double (&d)[6], const double (&e2)[5], double x, double &p, double &dp, double &ddp)
{
    double p0 = 1.0, q0 = 0.0, r0 = 0.0;                     // p_{k-2}, derivatives
    double p1 = d[0] - x, q1 = -1.0, r1 = 0.0;               // p_{k-1}
#pragma unroll
    for (int k = 1; k < 6; k++) {
        const double a = d[k] - x, b = e2[k - 1];
        const double p2 = a * p1 - b * p0;
        const double q2 = a * q1 - p1 - b * q0;
        const double r2 = a * r1 - 2.0 * q1 - b * r0;
        p0 = p1; q0 = q1; r0 = r1;
        p1 = p2; q1 = q2; r1 = r2;
    }
    p = p1; dp = q1; ddp = r1;
}
// one extreme root: dir = -1 starts left of the spectrum (lambda_min), +1 right of it (lambda_max)
NDT_HD double laguerre_extreme(const double (&d)[6], const double (&e2)[5], double x, int dir)
{
    for (int it = 0; it < 48; it++) {
        double p, dp, ddp;
        charpoly(d, e2, x, p, dp, ddp);
        if (p == 0.0) break;
        const double G = dp / p;
        const double Hh = G * G - ddp / p;
        double disc = 5.0 * (6.0 * Hh - G * G);
        if (!(disc > 0.0)) disc = 0.0;
        const double sq = sqrt(disc);
        const double den = (G >= 0.0) ? G + sq : G - sq;
        if (den == 0.0) break;
        const double a = 6.0 / den;
        // from outside the spectrum the step points inwards (a < 0 on the left, a > 0 on the right); a step the
        // other way is rounding noise at the root
        if ((dir < 0) ? !(a < 0.0) : !(a > 0.0)) break;
        const double xn = x - a;
        if (xn == x) break;
        const bool small = fabs(a) <= 4.0e-16 * (fabs(xn) > 1.0 ? fabs(xn) : 1.0);
        x = xn;
        if (small) break;
    }
    return x;
}
template <int K>
NDT_HD void householder_step(double (&a)[6][6])
{
    double sig = 0.0;
#pragma unroll
    for (int i = K + 2; i < 6; i++) sig += a[i][K] * a[i][K];
    if (sig > 0.0) {
        const double x0 = a[K + 1][K];
        const double nrm = sqrt(x0 * x0 + sig);
        const double alpha = (x0 > 0.0) ? -nrm : nrm;
        double v[6];
#pragma unroll
        for (int i = 0; i < 6; i++) v[i] = (i > K + 1) ? a[i][K] : 0.0;
        v[K + 1] = x0 - alpha;
        const double vtv = v[K + 1] * v[K + 1] + sig;
        const double beta = 2.0 / vtv;
        double pv[6], w[6];
        double ptv = 0.0;
#pragma unroll
        for (int i = K + 1; i < 6; i++) {
            double s = 0.0;
#pragma unroll
            for (int j = K + 1; j < 6; j++) s += a[i][j] * v[j];
            pv[i] = beta * s;
            ptv += pv[i] * v[i];
        }
        const double kk = 0.5 * beta * ptv;
#pragma unroll
        for (int i = K + 1; i < 6; i++) w[i] = pv[i] - kk * v[i];
#pragma unroll
        for (int i = K + 1; i < 6; i++)
#pragma unroll
            for (int j = K + 1; j < 6; j++) a[i][j] -= v[i] * w[j] + w[i] * v[j];
        a[K + 1][K] = alpha;
        a[K][K + 1] = alpha;
#pragma unroll
        for (int i = K + 2; i < 6; i++) { a[i][K] = 0.0; a[K][i] = 0.0; }
    }
}
}  // namespace ndt_eig6

NDT_HD void sym6_extreme_eigs(const double (&H)[6][6], double &lmin, double &lmax)
{
    double a[6][6];
    double scale = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) {
            a[i][j] = 0.5 * (H[i][j] + H[j][i]);
            scale = fmax(scale, fabs(a[i][j]));
        }
    if (!(scale > 0.0) || !(scale < 1.0e300)) { lmin = 0.0; lmax = 0.0; return; }   // zero (or non-finite) matrix
    const double inv = 1.0 / scale;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) a[i][j] *= inv;
    // Householder: after step K column K is zero below the sub-diagonal (template steps: every index is a constant)
    ndt_eig6::householder_step<0>(a);
    ndt_eig6::householder_step<1>(a);
    ndt_eig6::householder_step<2>(a);
    ndt_eig6::householder_step<3>(a);
    double d[6], e2[5];
    double lo = 1.0e300, hi = -1.0e300;
#pragma unroll
    for (int i = 0; i < 6; i++) d[i] = a[i][i];
#pragma unroll
    for (int i = 0; i < 5; i++) e2[i] = a[i + 1][i] * a[i + 1][i];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const double r = ((i > 0) ? fabs(a[i][i - 1]) : 0.0) + ((i < 5) ? fabs(a[i + 1][i]) : 0.0);
        lo = fmin(lo, d[i] - r);
        hi = fmax(hi, d[i] + r);
    }
    const double margin = 1.0e-3 * (hi - lo) + 1.0e-12;       // strictly outside the spectrum
    lmin = scale * ndt_eig6::laguerre_extreme(d, e2, lo - margin, -1);
    lmax = scale * ndt_eig6::laguerre_extreme(d, e2, hi + margin, +1);
}

// true when the symmetric matrix is positive definite (unpivoted Cholesky, registers only).  The factor is
// returned for chol_solve: l = L below the diagonal, dinv[j] = 1 / L[j][j].
template <int N>
NDT_HD bool chol_is_pd(const double (&A)[N][N], double (&l)[N][N], double (&dinv)[N])
{
    bool pd = true;
#pragma unroll
    for (int j = 0; j < N; j++) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= l[j][k] * l[j][k];
        pd = pd && (d > 0.0);
#if defined(__HIP_DEVICE_COMPILE__)
        double inv = rsqrt(d > 0.0 ? d : 1.0);       // v_rsq_f64 + refinement: a quarter of the sqrt + division chain
#else
        double inv = 1.0 / sqrt(d > 0.0 ? d : 1.0);
#endif
        l[j][j] = d * inv;
        dinv[j] = inv;
#pragma unroll
        for (int i = j + 1; i < N; i++) {
            double s = A[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= l[i][k] * l[j][k];
            l[i][j] = s * inv;
        }
    }
    return pd;
}

// x = A^-1 b from the Cholesky factor of a positive definite A (forward + back substitution)
template <int N>
NDT_HD void chol_solve(const double (&l)[N][N], const double (&dinv)[N], const double (&b)[N], double (&x)[N])
{
    double y[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= l[i][k] * y[k];
        y[i] = s * dinv[i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < N; k++) s -= l[k][i] * x[k];
        x[i] = s * dinv[i];
    }
}

// x = A^-1 b by LDL^T with symmetric diagonal pivoting (Hessian.ldlt().solve, fusion.h:966);
// zero pivots give a zero component like Eigen's LDLT::solve.
template <int NMAX>
NDT_HD void ldlt_solve_ws(int n, const double *A, const double *b, double *x, double *a, double *y, int *perm);

template <int NMAX>
NDT_HD void ldlt_solve(int n, const double *A, const double *b, double *x)
{
    double a[NMAX * NMAX], y[NMAX];
    int perm[NMAX];
    ldlt_solve_ws<NMAX>(n, A, b, x, a, y, perm);
}

// workspace variant: a[NMAX*NMAX], y[NMAX], perm[NMAX] supplied by the caller (LDS on the device,
// so that the dynamically indexed arrays do not live in scratch memory)
template <int NMAX>
NDT_HD void ldlt_solve_ws(int n, const double *A, const double *b, double *x, double *a, double *y, int *perm)
{
    for (int i = 0; i < n; i++) {
        perm[i] = i;
        for (int j = 0; j < n; j++) a[i * NMAX + j] = 0.5 * (A[i * n + j] + A[j * n + i]);
    }
    for (int k = 0; k < n; k++) {
        int piv = k;
        double best = fabs(a[k * NMAX + k]);
        for (int i = k + 1; i < n; i++)
            if (fabs(a[i * NMAX + i]) > best) { best = fabs(a[i * NMAX + i]); piv = i; }
        if (piv != k) {
            for (int j = 0; j < n; j++) { double t = a[k * NMAX + j]; a[k * NMAX + j] = a[piv * NMAX + j]; a[piv * NMAX + j] = t; }
            for (int i = 0; i < n; i++) { double t = a[i * NMAX + k]; a[i * NMAX + k] = a[i * NMAX + piv]; a[i * NMAX + piv] = t; }
            int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
        }
        double d = a[k * NMAX + k];
        if (fabs(d) <= 2.2250738585072014e-308) continue;
        for (int i = k + 1; i < n; i++) {
            double l = a[i * NMAX + k] / d;
            for (int j = k + 1; j < n; j++) a[i * NMAX + j] -= l * a[k * NMAX + j];
            a[i * NMAX + k] = l;
        }
        for (int j = k + 1; j < n; j++) a[k * NMAX + j] = 0.0;
    }
    for (int i = 0; i < n; i++) y[i] = b[perm[i]];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++) y[i] -= a[i * NMAX + j] * y[j];
    for (int i = 0; i < n; i++) {
        double d = a[i * NMAX + i];
        y[i] = (fabs(d) > 2.2250738585072014e-308) ? y[i] / d : 0.0;
    }
    for (int i = n - 1; i >= 0; i--)
        for (int j = i + 1; j < n; j++) y[i] -= a[j * NMAX + i] * y[j];
    for (int i = 0; i < n; i++) x[perm[i]] = y[i];
}

// 6x6 version of ldlt_solve with every index a compile-time constant (the matrix lives in registers):
// the data-dependent pivot only selects which statically indexed swap runs.  Same pivot rule (largest
// |diagonal| of the remaining block, first one on ties), same elimination and substitution formulas,
// hence the same arithmetic as ldlt_solve / Eigen's LDLT::solve on a 6x6 system.  Only the lower triangle is
// stored (21 values): the elimination of the full matrix reads its upper row K, which is the not yet scaled column K
// by symmetry, and the substitutions only ever read the lower triangle -- same products, same results, a third of
// the registers.
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): loops whose indices are constants from the start.
// (With `#pragma unroll` the indices only become constants after the unroller has run; by then the optimiser may have
// merged the branches of a pivot switch into indexed accesses, and the register array goes to the stack.)
template <int... Is, class F>
NDT_HD void ndt_static_for_impl(std::integer_sequence<int, Is...>, F &&f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
NDT_HD void ndt_static_for(F &&f) { ndt_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

namespace ndt_ldlt6 {
NDT_HD constexpr int tri(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }
template <int K, int C>
NDT_HD void swap_kc(double (&a)[21], double (&y)[6])
{
    // symmetric exchange of the indices K < C (rows and columns)
    ndt_static_for<6>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        if constexpr (i != K && i != C) { double t = a[tri(i, K)]; a[tri(i, K)] = a[tri(i, C)]; a[tri(i, C)] = t; }
    });
    double t = a[tri(K, K)]; a[tri(K, K)] = a[tri(C, C)]; a[tri(C, C)] = t;
    t = y[K]; y[K] = y[C]; y[C] = t;
}
template <int K>
NDT_HD void swap_k(double (&a)[21], double (&y)[6], int piv)
{
    if constexpr (K < 1) { if (piv == 1) swap_kc<K, 1>(a, y); }
    if constexpr (K < 2) { if (piv == 2) swap_kc<K, 2>(a, y); }
    if constexpr (K < 3) { if (piv == 3) swap_kc<K, 3>(a, y); }
    if constexpr (K < 4) { if (piv == 4) swap_kc<K, 4>(a, y); }
    if constexpr (K < 5) { if (piv == 5) swap_kc<K, 5>(a, y); }
}
template <int K>
NDT_HD void unswap_k(double (&y)[6], int piv)
{
    // x[perm[i]] = y[i]: undo the row exchanges in reverse order
    ndt_static_for<6>([&](auto Cc) __attribute__((always_inline)) {
        constexpr int c = decltype(Cc)::value;
        if constexpr (c > K) { if (piv == c) { double t = y[K]; y[K] = y[c]; y[c] = t; } }
    });
}
template <int K>
NDT_HD void step(double (&a)[21], double (&y)[6], int (&pivs)[6])
{
    int piv = K;
    double best = fabs(a[tri(K, K)]);
    ndt_static_for<6>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        if constexpr (i > K) { if (fabs(a[tri(i, i)]) > best) { best = fabs(a[tri(i, i)]); piv = i; } }
    });
    pivs[K] = piv;
    if (piv != K) swap_k<K>(a, y, piv);
    const double d = a[tri(K, K)];
    if (fabs(d) > 2.2250738585072014e-308) {
        ndt_static_for<6>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            if constexpr (i > K) {
                // (rows in descending order: row i needs the unscaled column entries a(j, K), j <= i, and scales its own last)
                constexpr int r = 5 + (K + 1) - i;
                const double l = a[tri(r, K)] / d;
                ndt_static_for<6>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value;
                    if constexpr (j > K && j <= r) a[tri(r, j)] -= l * a[tri(j, K)];
                });
                a[tri(r, K)] = l;
            }
        });
    }
}
// the matrix as its lower triangle a[tri(i, j)], i >= j; y: right-hand side in, solution out
NDT_HD void solve_packed(double (&a)[21], double (&y)[6])
{
    int pivs[6];
    step<0>(a, y, pivs);
    step<1>(a, y, pivs);
    step<2>(a, y, pivs);
    step<3>(a, y, pivs);
    step<4>(a, y, pivs);
    pivs[5] = 5;
    ndt_static_for<6>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        ndt_static_for<6>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            if constexpr (j < i) y[i] -= a[tri(i, j)] * y[j];
        });
    });
    ndt_static_for<6>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        const double d = a[tri(i, i)];
        y[i] = (fabs(d) > 2.2250738585072014e-308) ? y[i] / d : 0.0;
    });
    ndt_static_for<6>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = 5 - decltype(I)::value;
        ndt_static_for<6>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            if constexpr (j > i) y[i] -= a[tri(j, i)] * y[j];
        });
    });
    unswap_k<4>(y, pivs[4]);
    unswap_k<3>(y, pivs[3]);
    unswap_k<2>(y, pivs[2]);
    unswap_k<1>(y, pivs[1]);
    unswap_k<0>(y, pivs[0]);
}
}  // namespace ndt_ldlt6

NDT_HD void ldlt_solve_static6(const double (&A)[6][6], const double (&b)[6], double (&x)[6])
{
    double a[21], y[6];
    ndt_static_for<6>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        y[i] = b[i];
        ndt_static_for<6>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            if constexpr (j <= i) a[ndt_ldlt6::tri(i, j)] = 0.5 * (A[i][j] + A[j][i]);
        });
    });
    ndt_ldlt6::solve_packed(a, y);
    ndt_static_for<6>([&](auto I) __attribute__((always_inline)) { x[decltype(I)::value] = y[decltype(I)::value]; });
}

// LazyGrid::getIndexForPoint: idx = floor((p - centre)/res + 0.5) + size/2.0, double -> int.
// Contraction is off so that the host oracle and the device agree bit-for-bit at cell faces.
// (the same with size / 2.0 handed in: NdtGrid::half, a kernel argument, i.e. a scalar register -- computed in the kernel it is a
//  wave-uniform value in a vector register, and the flat build kernel spilled three of them)
NDT_HD int lazygrid_index_half(double p, double centre, double res, double half)
{
#pragma clang fp contract(off)
    double v = floor((p - centre) / res + 0.5) + half;
    if (!(v > -2.0e9 && v < 2.0e9)) return -1;
    return (int)v;
}
NDT_HD int lazygrid_index(double p, double centre, double res, int size)
{
#pragma clang fp contract(off)
    double v = floor((p - centre) / res + 0.5) + size / 2.0;
    if (!(v > -2.0e9 && v < 2.0e9)) return -1;   // NaN / far away: outside any grid
    return (int)v;
}
