/*
 * ndt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See ndt_oracle.h for provenance ("PARITY UNPINNED") and the usage rule.
 *
 * Reference files restated (paths relative to /root/reference):
 *   [fusion.h]  ndt_feature/include/ndt_feature/ndt_matcher_d2d_fusion.h
 *   [fuser.cpp] ndt_feature/src/ndt_feature_src/ndt_feature_fuser_hmt.cpp
 *   [graph.cpp] ndt_feature/src/ndt_feature_src/ndt_feature_graph.cpp
 *   [debug.cpp] ndt_feature/src/ndt_odom_debug.cpp
 * plus the external perception_oru semantics summarised in SURVEY.md App. A.
 */
#include "ndt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* small dense algebra                                                  */
/* ------------------------------------------------------------------ */

typedef struct { double v[3]; } vec3;
typedef struct { double m[3][3]; } mat3;

static mat3 m3_zero(void) { mat3 r; memset(&r, 0, sizeof r); return r; }
static mat3 m3_mul(mat3 a, mat3 b)
{
    mat3 r = m3_zero();
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) r.m[i][j] += a.m[i][k] * b.m[k][j];
    return r;
}
static mat3 m3_add(mat3 a, mat3 b)
{
    mat3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] + b.m[i][j];
    return r;
}
static mat3 m3_T(mat3 a)
{
    mat3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
    return r;
}
static vec3 m3_v(mat3 a, vec3 x)
{
    vec3 r;
    for (int i = 0; i < 3; i++) r.v[i] = a.m[i][0] * x.v[0] + a.m[i][1] * x.v[1] + a.m[i][2] * x.v[2];
    return r;
}
static double v3_dot(vec3 a, vec3 b) { return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2]; }

/* [e_k]x */
static mat3 cross_mat(int k)
{
    mat3 r = m3_zero();
    if (k == 0) { r.m[1][2] = -1; r.m[2][1] = 1; }
    if (k == 1) { r.m[0][2] = 1; r.m[2][0] = -1; }
    if (k == 2) { r.m[0][1] = -1; r.m[1][0] = 1; }
    return r;
}

/* Matrix3d::computeInverseAndDetWithCheck (Eigen default threshold:
 * |det| > dummy_precision = 1e-12).  Used on CSum in derivativesNDT. */
static int m3_inverse_check(mat3 a, mat3 *inv, double *det_out)
{
    double c00 = a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1];
    double c01 = a.m[1][2] * a.m[2][0] - a.m[1][0] * a.m[2][2];
    double c02 = a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0];
    double det = a.m[0][0] * c00 + a.m[0][1] * c01 + a.m[0][2] * c02;
    *det_out = det;
    if (!(fabs(det) > 1e-12)) return 0;
    double id = 1.0 / det;
    inv->m[0][0] = c00 * id;
    inv->m[1][0] = c01 * id;
    inv->m[2][0] = c02 * id;
    inv->m[0][1] = (a.m[0][2] * a.m[2][1] - a.m[0][1] * a.m[2][2]) * id;
    inv->m[1][1] = (a.m[0][0] * a.m[2][2] - a.m[0][2] * a.m[2][0]) * id;
    inv->m[2][1] = (a.m[0][1] * a.m[2][0] - a.m[0][0] * a.m[2][1]) * id;
    inv->m[0][2] = (a.m[0][1] * a.m[1][2] - a.m[0][2] * a.m[1][1]) * id;
    inv->m[1][2] = (a.m[0][2] * a.m[1][0] - a.m[0][0] * a.m[1][2]) * id;
    inv->m[2][2] = (a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0]) * id;
    return 1;
}

/* cyclic Jacobi, n <= 6, row-major A (symmetric); evals ascending, evecs
 * columns (row-major n x n).  Stands in for Eigen::SelfAdjointEigenSolver
 * ([fusion.h]:922-928 on the Hessian; NDTCell::rescaleCovariance on 3x3). */
int oracle_eig_sym(int n, const double *A, double *evals, double *evecs)
{
    double a[6][6], v[6][6];
    if (n < 1 || n > 6) return -1;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            a[i][j] = 0.5 * (A[i * n + j] + A[j * n + i]);
            v[i][j] = (i == j);
        }
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; i++) {
            diag += a[i][i] * a[i][i];
            for (int j = i + 1; j < n; j++) off += a[i][j] * a[i][j];
        }
        if (off == 0.0 || off <= 1e-60 * diag) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                if (a[p][q] == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int order[6];
    for (int i = 0; i < n; i++) order[i] = i;
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++)
            if (a[order[j]][order[j]] < a[order[i]][order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
    for (int i = 0; i < n; i++) {
        evals[i] = a[order[i]][order[i]];
        if (evecs)
            for (int k = 0; k < n; k++) evecs[k * n + i] = v[k][order[i]];
    }
    return 0;
}

/* Hessian.ldlt().solve(b) ([fusion.h]:966): LDL^T with symmetric diagonal
 * pivoting (largest |diagonal|, as Eigen::LDLT), zero pivots give a zero
 * component (Eigen's pseudo-inverse convention). */
int oracle_ldlt_solve(int n, const double *A, const double *b, double *x)
{
    double a[6][6], y[6];
    int perm[6];
    if (n < 1 || n > 6) return -1;
    for (int i = 0; i < n; i++) {
        perm[i] = i;
        for (int j = 0; j < n; j++) a[i][j] = 0.5 * (A[i * n + j] + A[j * n + i]);
    }
    for (int k = 0; k < n; k++) {
        int piv = k;
        double best = fabs(a[k][k]);
        for (int i = k + 1; i < n; i++)
            if (fabs(a[i][i]) > best) { best = fabs(a[i][i]); piv = i; }
        if (piv != k) {
            for (int j = 0; j < n; j++) { double t = a[k][j]; a[k][j] = a[piv][j]; a[piv][j] = t; }
            for (int i = 0; i < n; i++) { double t = a[i][k]; a[i][k] = a[i][piv]; a[i][piv] = t; }
            int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
        }
        double d = a[k][k];
        if (fabs(d) <= DBL_MIN) continue;
        for (int i = k + 1; i < n; i++) {
            double l = a[i][k] / d;
            for (int j = k + 1; j < n; j++) a[i][j] -= l * a[k][j];
            a[i][k] = l;
        }
        for (int j = k + 1; j < n; j++) a[k][j] = 0.0; /* keep strictly-lower L and diagonal D */
    }
    for (int i = 0; i < n; i++) y[i] = b[perm[i]];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++) y[i] -= a[i][j] * y[j];
    for (int i = 0; i < n; i++) y[i] = (fabs(a[i][i]) > DBL_MIN) ? y[i] / a[i][i] : 0.0;
    for (int i = n - 1; i >= 0; i--)
        for (int j = i + 1; j < n; j++) y[i] -= a[j][i] * y[j];
    for (int i = 0; i < n; i++) x[perm[i]] = y[i];
    return 0;
}

/* TR = Translation(p0,p1,p2) * Rx(p3) * Ry(p4) * Rz(p5)   [fusion.h]:1036-1039
 * 4x4 column-major like Eigen::Affine3d::data(). */
void oracle_pose_to_T(const double p[6], double T[16])
{
    double cx = cos(p[3]), sx = sin(p[3]);
    double cy = cos(p[4]), sy = sin(p[4]);
    double cz = cos(p[5]), sz = sin(p[5]);
    mat3 Rx = {{{1, 0, 0}, {0, cx, -sx}, {0, sx, cx}}};
    mat3 Ry = {{{cy, 0, sy}, {0, 1, 0}, {-sy, 0, cy}}};
    mat3 Rz = {{{cz, -sz, 0}, {sz, cz, 0}, {0, 0, 1}}};
    mat3 R = m3_mul(m3_mul(Rx, Ry), Rz);
    memset(T, 0, 16 * sizeof(double));
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) T[c * 4 + r] = R.m[r][c];
    T[12] = p[0];
    T[13] = p[1];
    T[14] = p[2];
    T[15] = 1.0;
}

static void T_mul(const double A[16], const double B[16], double C[16])
{
    double r[16];
    for (int c = 0; c < 4; c++)
        for (int rr = 0; rr < 4; rr++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += A[k * 4 + rr] * B[c * 4 + k];
            r[c * 4 + rr] = s;
        }
    memcpy(C, r, sizeof r);
}

static mat3 T_rot(const double T[16])
{
    mat3 R;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) R.m[r][c] = T[c * 4 + r];
    return R;
}

/* computeScore/Gradient/HessianMahalanobis  [fusion.h]:11-32 */
double oracle_mahalanobis(const double x[6], const double Q[36], double g[6], double H[36])
{
    double s = 0;
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            H[i * 6 + j] = Q[j * 6 + i] + Q[i * 6 + j];
            s += x[i] * Q[i * 6 + j] * x[j];
        }
    for (int i = 0; i < 6; i++) {
        g[i] = 0;
        for (int j = 0; j < 6; j++) g[i] += H[i * 6 + j] * x[j];
    }
    return s;
}

/* ------------------------------------------------------------------ */
/* LazyGrid / NDTMap                                                    */
/* ------------------------------------------------------------------ */

typedef struct {
    double mean[3];
    mat3 cov;
    int n;
    int has_gaussian;
    int idx[3];
} ocell;

struct oracle_map {
    double res;
    double centre[3];
    int size[3];
    size_t nslots;
    int32_t *cell_of_slot; /* -1 = NULL pointer in dataArray */
    ocell *cells;
    size_t ncells, capcells;
    /* points waiting for computeNDTCells, grouped per cell (NDTCell::points_) */
    float *pts;       /* xyz packed, sorted by cell, insertion order kept */
    size_t *pt_begin; /* ncells+1 */
};

oracle_map *oracle_map_create(double res, const double centre[3], const double size_m[3])
{
    oracle_map *m = (oracle_map *)calloc(1, sizeof *m);
    if (!m) return NULL;
    m->res = res;
    m->nslots = 1;
    for (int a = 0; a < 3; a++) {
        m->centre[a] = centre[a];
        /* LazyGrid::initialize: sizeX = abs(ceil(sizeXmeters / cellSizeX)) */
        m->size[a] = abs((int)ceil(size_m[a] / res));
        m->nslots *= (size_t)m->size[a];
    }
    m->cell_of_slot = (int32_t *)malloc(m->nslots * sizeof(int32_t));
    if (!m->cell_of_slot) { free(m); return NULL; }
    for (size_t i = 0; i < m->nslots; i++) m->cell_of_slot[i] = -1;
    return m;
}

static void map_clear(oracle_map *m)
{
    for (size_t i = 0; i < m->nslots; i++) m->cell_of_slot[i] = -1;
    free(m->cells); m->cells = NULL; m->ncells = m->capcells = 0;
    free(m->pts); m->pts = NULL;
    free(m->pt_begin); m->pt_begin = NULL;
}

void oracle_map_destroy(oracle_map *m)
{
    if (!m) return;
    map_clear(m);
    free(m->cell_of_slot);
    free(m);
}

/* LazyGrid::getIndexForPoint:
 *   indX = floor((p.x - centerX)/cellSizeX + 0.5) + sizeX/2.0;   (double -> int)  */
static void index_for_point(const oracle_map *m, const double p[3], int idx[3])
{
    for (int a = 0; a < 3; a++)
        idx[a] = (int)(floor((p[a] - m->centre[a]) / m->res + 0.5) + m->size[a] / 2.0);
}
static int idx_inside(const oracle_map *m, const int idx[3])
{
    for (int a = 0; a < 3; a++)
        if (idx[a] < 0 || idx[a] >= m->size[a]) return 0;
    return 1;
}
static size_t slot_of(const oracle_map *m, const int idx[3])
{
    return ((size_t)idx[0] * m->size[1] + idx[1]) * m->size[2] + idx[2];
}

int oracle_map_index_for_point(const oracle_map *m, const double p[3], int idx[3])
{
    index_for_point(m, p, idx);
    return idx_inside(m, idx);
}

static int cell_for_slot_create(oracle_map *m, size_t slot, const int idx[3])
{
    int32_t c = m->cell_of_slot[slot];
    if (c >= 0) return c;
    if (m->ncells == m->capcells) {
        size_t nc = m->capcells ? 2 * m->capcells : 256;
        ocell *p = (ocell *)realloc(m->cells, nc * sizeof(ocell));
        if (!p) return -1;
        m->cells = p;
        m->capcells = nc;
    }
    ocell *ce = &m->cells[m->ncells];
    memset(ce, 0, sizeof *ce);
    ce->idx[0] = idx[0]; ce->idx[1] = idx[1]; ce->idx[2] = idx[2];
    m->cell_of_slot[slot] = (int32_t)m->ncells;
    return (int)m->ncells++;
}

int oracle_map_load_points(oracle_map *m, const float *xyz, size_t n, size_t stride, double range_limit,
                           const double *range_origin)
{
    map_clear(m);
    int32_t *cell_of_pt = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
    if (!cell_of_pt) return -1;
    /* NDTMap::loadPointCloud: skip NaN, skip ||p|| > range_limit, LazyGrid::addPoint drops
     * points whose index falls outside the grid. */
    for (size_t i = 0; i < n; i++) {
        const float *q = xyz + i * stride;
        cell_of_pt[i] = -1;
        if (isnan(q[0]) || isnan(q[1]) || isnan(q[2])) continue;
        double p[3] = {q[0], q[1], q[2]};
        if (range_limit > 0) {
            double d[3] = {p[0], p[1], p[2]};
            if (range_origin)
                for (int a = 0; a < 3; a++) d[a] -= range_origin[a];
            if (sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) > range_limit) continue;
        }
        int idx[3];
        index_for_point(m, p, idx);
        if (!idx_inside(m, idx)) continue;
        int c = cell_for_slot_create(m, slot_of(m, idx), idx);
        if (c < 0) { free(cell_of_pt); return -1; }
        cell_of_pt[i] = c;
        m->cells[c].n++;
    }
    m->pt_begin = (size_t *)calloc(m->ncells + 1, sizeof(size_t));
    for (size_t c = 0; c < m->ncells; c++) m->pt_begin[c + 1] = m->pt_begin[c] + (size_t)m->cells[c].n;
    size_t tot = m->pt_begin[m->ncells];
    m->pts = (float *)malloc((tot ? tot : 1) * 3 * sizeof(float));
    size_t *fill = (size_t *)calloc(m->ncells ? m->ncells : 1, sizeof(size_t));
    for (size_t i = 0; i < n; i++) {
        int c = cell_of_pt[i];
        if (c < 0) continue;
        size_t o = (m->pt_begin[c] + fill[c]++) * 3;
        const float *q = xyz + i * stride;
        m->pts[o] = q[0]; m->pts[o + 1] = q[1]; m->pts[o + 2] = q[2];
    }
    free(fill);
    free(cell_of_pt);
    return 0;
}

#define ORACLE_DEGENERATE_REL 1e-9

/* NDTCell::rescaleCovariance */
static void rescale_covariance(ocell *ce, double eval_factor)
{
    double A[9], ev[3], V[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) A[i * 3 + j] = ce->cov.m[i][j];
    oracle_eig_sym(3, A, ev, V);
    /* Upstream tests evals <= 0.  A rank-deficient sample covariance (3 points, collinear or
     * exactly coplanar points) has an exact zero eigenvalue that floating point turns into
     * +-1e-17*lambda_max, i.e. a coin flip in the reference itself.  Oracle and HIP path both
     * resolve it the exact-arithmetic way: lambda_min <= 1e-9*lambda_max counts as "<= 0". */
    if (ev[2] <= 0 || ev[0] <= ORACLE_DEGENERATE_REL * ev[2]) {
        ce->has_gaussian = 0;
        return;
    }
    ce->has_gaussian = 1;
    double mx = ev[2];
    int recalc = 0;
    for (int k = 0; k < 3; k++)
        if (mx > ev[k] * eval_factor) { ev[k] = mx / eval_factor; recalc = 1; }
    if (recalc)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += V[i * 3 + k] * ev[k] * V[j * 3 + k];
                ce->cov.m[i][j] = s;
            }
}

/* NDTCell::computeGaussian, first-Gaussian branch:
 * occupancy: logodd = n*log(0.6/0.4) > 0 for any cell that holds points, so
 * occ > 0; points_.size() < n_min -> no Gaussian; mean = sum/n;
 * cov = sum (p-mean)(p-mean)^T / (n-1); rescaleCovariance. */
int oracle_map_compute_cells(oracle_map *m, int n_min, double eval_factor)
{
    if (!m->pt_begin) return 0;
    for (size_t c = 0; c < m->ncells; c++) {
        ocell *ce = &m->cells[c];
        size_t b = m->pt_begin[c], e = m->pt_begin[c + 1], n = e - b;
        ce->has_gaussian = 0;
        if ((int)n < n_min || n == 0) continue;
        double mean[3] = {0, 0, 0};
        for (size_t i = b; i < e; i++)
            for (int a = 0; a < 3; a++) mean[a] += (double)m->pts[i * 3 + a];
        for (int a = 0; a < 3; a++) mean[a] /= (double)n;
        mat3 S = m3_zero();
        for (size_t i = b; i < e; i++) {
            double d[3];
            for (int a = 0; a < 3; a++) d[a] = (double)m->pts[i * 3 + a] - mean[a];
            for (int r = 0; r < 3; r++)
                for (int q = 0; q < 3; q++) S.m[r][q] += d[r] * d[q];
        }
        for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++) ce->cov.m[r][q] = S.m[r][q] / (double)(n - 1);
        for (int a = 0; a < 3; a++) ce->mean[a] = mean[a];
        rescale_covariance(ce, eval_factor);
    }
    free(m->pts); m->pts = NULL;
    free(m->pt_begin); m->pt_begin = NULL;
    return 0;
}

int oracle_map_num_cells(const oracle_map *m)
{
    int k = 0;
    for (size_t c = 0; c < m->ncells; c++) k += m->cells[c].has_gaussian;
    return k;
}

int oracle_map_export_cells(const oracle_map *m, double *mean3, double *cov9, int32_t *idx3, int32_t *npts)
{
    int k = 0;
    for (size_t s = 0; s < m->nslots; s++) {
        int32_t c = m->cell_of_slot[s];
        if (c < 0 || !m->cells[c].has_gaussian) continue;
        const ocell *ce = &m->cells[c];
        for (int a = 0; a < 3; a++) {
            if (mean3) mean3[3 * k + a] = ce->mean[a];
            if (idx3) idx3[3 * k + a] = ce->idx[a];
            for (int b = 0; b < 3; b++)
                if (cov9) cov9[9 * k + 3 * a + b] = ce->cov.m[a][b];
        }
        if (npts) npts[k] = ce->n;
        k++;
    }
    return k;
}

int oracle_map_set_cells(oracle_map *m, const double *mean3, const double *cov9, size_t ncells)
{
    map_clear(m);
    for (size_t i = 0; i < ncells; i++) {
        int idx[3];
        index_for_point(m, mean3 + 3 * i, idx);
        if (!idx_inside(m, idx)) continue;
        int c = cell_for_slot_create(m, slot_of(m, idx), idx);
        if (c < 0) return -1;
        ocell *ce = &m->cells[c];
        for (int a = 0; a < 3; a++) {
            ce->mean[a] = mean3[3 * i + a];
            for (int b = 0; b < 3; b++) ce->cov.m[a][b] = cov9[9 * i + 3 * a + b];
        }
        ce->n = 1;
        ce->has_gaussian = 1;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* NDTMatcherD2D::derivativesNDT (+ computeDerivativesLocal,            */
/* updateGradientHessianLocal), SURVEY App. A.4                         */
/* ------------------------------------------------------------------ */

typedef struct {
    vec3 J[6];      /* _Jest columns */
    mat3 Z[6];      /* _Zest blocks (zero for a<3) */
    vec3 Hm[6][6];  /* _Hest blocks */
    mat3 ZH[6][6];  /* _ZHest blocks */
} local_derivs;

static void compute_derivatives_local(vec3 m, mat3 C, int with_hessian, local_derivs *L)
{
    memset(L, 0, sizeof *L);
    for (int a = 0; a < 3; a++) L->J[a].v[a] = 1.0;
    mat3 E[3];
    for (int k = 0; k < 3; k++) {
        E[k] = cross_mat(k);
        L->J[3 + k] = m3_v(E[k], m);                             /* e_k x m */
        L->Z[3 + k] = m3_add(m3_mul(E[k], C), m3_mul(C, m3_T(E[k])));
    }
    if (!with_hessian) return;
    for (int a = 0; a < 3; a++)
        for (int b = a; b < 3; b++) {
            mat3 AB = m3_mul(E[a], E[b]);
            vec3 h = m3_v(AB, m);                                 /* e_a x (e_b x m) */
            mat3 zh = m3_add(m3_add(m3_mul(AB, C), m3_mul(C, m3_T(AB))),
                             m3_add(m3_mul(m3_mul(E[a], C), m3_T(E[b])),
                                    m3_mul(m3_mul(E[b], C), m3_T(E[a]))));
            L->Hm[3 + a][3 + b] = h;
            L->Hm[3 + b][3 + a] = h;
            L->ZH[3 + a][3 + b] = zh;
            L->ZH[3 + b][3 + a] = zh;
        }
}

static void update_gradient_hessian_local(double g[6], double H[36], vec3 x, mat3 B, double sh,
                                          const local_derivs *L, int with_hessian, double lfd2)
{
    vec3 xB = m3_v(m3_T(B), x); /* x^T B */
    double Q[6], xtBJ[6], xtBZBx[6];
    vec3 xBZB[6];               /* x^T B Z_a B */
    for (int a = 0; a < 6; a++) {
        xtBJ[a] = v3_dot(xB, L->J[a]);
        vec3 t = m3_v(m3_T(L->Z[a]), xB);    /* (x^T B Z_a)^T */
        xBZB[a] = m3_v(m3_T(B), t);
        xtBZBx[a] = v3_dot(xBZB[a], x);
        Q[a] = 2.0 * xtBJ[a] - xtBZBx[a];
    }
    double factor = -(lfd2 / 2.0) * sh;
    for (int a = 0; a < 6; a++) g[a] += factor * Q[a];
    if (!with_hessian) return;
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) {
            double JtBJ = v3_dot(m3_v(B, L->J[a]), L->J[b]);
            double xtBH = v3_dot(xB, L->Hm[a][b]);
            double xtBZBJ_ab = v3_dot(xBZB[a], L->J[b]);
            double xtBZBJ_ba = v3_dot(xBZB[b], L->J[a]);
            vec3 t = m3_v(m3_T(L->Z[b]), xBZB[a]);     /* (x^T B Z_a B Z_b)^T */
            double xtBZBZBx_ab = v3_dot(m3_v(m3_T(B), t), x);
            vec3 t2 = m3_v(m3_T(L->Z[a]), xBZB[b]);
            double xtBZBZBx_ba = v3_dot(m3_v(m3_T(B), t2), x);
            vec3 t3 = m3_v(m3_T(L->ZH[a][b]), xB);
            double xtBZhBx = v3_dot(m3_v(m3_T(B), t3), x);
            double d2q = 2.0 * JtBJ + 2.0 * xtBH - xtBZhBx - 2.0 * xtBZBJ_ab - 2.0 * xtBZBJ_ba +
                         xtBZBZBx_ab + xtBZBZBx_ba;
            H[a * 6 + b] += factor * (d2q - (lfd2 / 2.0) * Q[a] * Q[b]);
        }
}

/* LazyGrid::getClosestNDTCells order: offsets 0,+1,-1,+2,-2 per axis, x outer, z inner */
static int nb_offset(int k) { return (k % 2 == 0) ? k / 2 : -(k / 2); }

static double derivatives_cells(const oracle_map *target, const ocell *src, size_t msrc, int n_neighbours,
                                int with_hessian, double lfd1, double lfd2, double g[6], double H[36])
{
    double score = 0;
    memset(g, 0, 6 * sizeof(double));
    memset(H, 0, 36 * sizeof(double));
    for (size_t i = 0; i < msrc; i++) {
        vec3 mm = {{src[i].mean[0], src[i].mean[1], src[i].mean[2]}};
        mat3 CM = src[i].cov;
        local_derivs L;
        compute_derivatives_local(mm, CM, with_hessian, &L);
        int ic[3];
        index_for_point(target, src[i].mean, ic);
        for (int kx = 1; kx < 2 * n_neighbours + 2; kx++)
            for (int ky = 1; ky < 2 * n_neighbours + 2; ky++)
                for (int kz = 1; kz < 2 * n_neighbours + 2; kz++) {
                    int idx[3] = {ic[0] + nb_offset(kx), ic[1] + nb_offset(ky), ic[2] + nb_offset(kz)};
                    if (!idx_inside(target, idx)) continue;
                    int32_t c = target->cell_of_slot[slot_of(target, idx)];
                    if (c < 0 || !target->cells[c].has_gaussian) continue;
                    const ocell *tc = &target->cells[c];
                    vec3 x = {{mm.v[0] - tc->mean[0], mm.v[1] - tc->mean[1], mm.v[2] - tc->mean[2]}};
                    mat3 CS = m3_add(tc->cov, CM), B;
                    double det;
                    if (!m3_inverse_check(CS, &B, &det)) continue;
                    double l = v3_dot(x, m3_v(B, x));
                    if (l * 0 != 0) continue;
                    double sh = -lfd1 * exp(-lfd2 * l / 2.0);
                    update_gradient_hessian_local(g, H, x, B, sh, &L, with_hessian, lfd2);
                    score += sh;
                }
    }
    return score;
}

double oracle_derivatives(const oracle_map *target, const double *src_mean3, const double *src_cov9, size_t m,
                          int n_neighbours, int compute_hessian, double lfd1, double lfd2, double g[6],
                          double H[36])
{
    ocell *src = (ocell *)calloc(m ? m : 1, sizeof(ocell));
    for (size_t i = 0; i < m; i++) {
        for (int a = 0; a < 3; a++) {
            src[i].mean[a] = src_mean3[3 * i + a];
            for (int b = 0; b < 3; b++) src[i].cov.m[a][b] = src_cov9[9 * i + 3 * a + b];
        }
        src[i].has_gaussian = 1;
    }
    double s = derivatives_cells(target, src, m, n_neighbours, compute_hessian, lfd1, lfd2, g, H);
    free(src);
    return s;
}

/* NDTMap::pseudoTransformNDT(T): copies of all Gaussian cells, mean' = T mean,
 * cov' = R cov R^T  ([fusion.h]:840).  Order: slot order. */
static ocell *pseudo_transform(const oracle_map *m, const double T[16], size_t *n_out)
{
    size_t k = 0;
    ocell *out = (ocell *)calloc(m->ncells ? m->ncells : 1, sizeof(ocell));
    mat3 R = T_rot(T);
    for (size_t s = 0; s < m->nslots; s++) {
        int32_t c = m->cell_of_slot[s];
        if (c < 0 || !m->cells[c].has_gaussian) continue;
        out[k] = m->cells[c];
        vec3 mu = {{m->cells[c].mean[0], m->cells[c].mean[1], m->cells[c].mean[2]}};
        vec3 r = m3_v(R, mu);
        for (int a = 0; a < 3; a++) out[k].mean[a] = r.v[a] + T[12 + a];
        out[k].cov = m3_mul(m3_mul(R, m->cells[c].cov), m3_T(R));
        k++;
    }
    *n_out = k;
    return out;
}

static void transform_cells(const ocell *in, ocell *out, size_t n, const double T[16])
{
    mat3 R = T_rot(T);
    for (size_t i = 0; i < n; i++) {
        vec3 mu = {{in[i].mean[0], in[i].mean[1], in[i].mean[2]}};
        vec3 r = m3_v(R, mu);
        mat3 C = m3_mul(m3_mul(R, in[i].cov), m3_T(R));
        out[i] = in[i];
        for (int a = 0; a < 3; a++) out[i].mean[a] = r.v[a] + T[12 + a];
        out[i].cov = C;
    }
}

double oracle_score_at(const oracle_map *target, const oracle_map *source, const double T[16], const double p[6],
                       int n_neighbours, double lfd1, double lfd2)
{
    size_t n;
    double TR[16], TT[16], g[6], H[36];
    oracle_pose_to_T(p, TR);
    T_mul(TR, T, TT);
    ocell *cells = pseudo_transform(source, TT, &n);
    double s = derivatives_cells(target, cells, n, n_neighbours, 0, lfd1, lfd2, g, H);
    free(cells);
    return s;
}

/* ------------------------------------------------------------------ */
/* More-Thuente                                                         */
/* ------------------------------------------------------------------ */

static double dmin(double a, double b) { return a < b ? a : b; }
static double dmax(double a, double b) { return a > b ? a : b; }
static double absmax3(double a, double b, double c) { return dmax(dmax(fabs(a), fabs(b)), fabs(c)); }

/* MoreThuente::cstep == MINPACK-2 dcstep (More & Thuente 1994, "Line search
 * algorithms with guaranteed sufficient decrease", ACM TOMS 20(3)).  Called at
 * [fusion.h]:756,775 with (stx,fx,dgx,sty,fy,dgy,stp,f,dg,brackt,stmin,stmax). */
int oracle_mt_cstep(double *stx, double *fx, double *dx, double *sty, double *fy, double *dy, double *stp,
                    double fp, double dp, int *brackt, double stmin, double stmax)
{
    int info = 0, bound;
    double theta, s, gamma, p, q, r, stpc, stpq, stpf;

    if ((*brackt && ((*stp <= dmin(*stx, *sty)) || (*stp >= dmax(*stx, *sty)))) ||
        (*dx * (*stp - *stx) >= 0.0) || (stmax < stmin))
        return info;

    double sgnd = dp * (*dx / fabs(*dx));

    if (fp > *fx) {
        /* case 1: higher function value -> minimum bracketed */
        info = 1;
        bound = 1;
        theta = 3 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = absmax3(theta, *dx, dp);
        gamma = s * sqrt(((theta / s) * (theta / s)) - (*dx / s) * (dp / s));
        if (*stp < *stx) gamma = -gamma;
        p = (gamma - *dx) + theta;
        q = ((gamma - *dx) + gamma) + dp;
        r = p / q;
        stpc = *stx + r * (*stp - *stx);
        stpq = *stx + ((*dx / ((*fx - fp) / (*stp - *stx) + *dx)) / 2) * (*stp - *stx);
        if (fabs(stpc - *stx) < fabs(stpq - *stx)) stpf = stpc;
        else stpf = stpc + (stpq - stpc) / 2;
        *brackt = 1;
    } else if (sgnd < 0.0) {
        /* case 2: lower value, derivatives of opposite sign -> bracketed */
        info = 2;
        bound = 0;
        theta = 3 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = absmax3(theta, *dx, dp);
        gamma = s * sqrt(((theta / s) * (theta / s)) - (*dx / s) * (dp / s));
        if (*stp > *stx) gamma = -gamma;
        p = (gamma - dp) + theta;
        q = ((gamma - dp) + gamma) + *dx;
        r = p / q;
        stpc = *stp + r * (*stx - *stp);
        stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
        if (fabs(stpc - *stp) > fabs(stpq - *stp)) stpf = stpc;
        else stpf = stpq;
        *brackt = 1;
    } else if (fabs(dp) < fabs(*dx)) {
        /* case 3: lower value, same sign, derivative magnitude decreases */
        info = 3;
        bound = 1;
        theta = 3 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = absmax3(theta, *dx, dp);
        gamma = s * sqrt(dmax(0, (theta / s) * (theta / s) - (*dx / s) * (dp / s)));
        if (*stp > *stx) gamma = -gamma;
        p = (gamma - dp) + theta;
        q = (gamma + (*dx - dp)) + gamma;
        r = p / q;
        if ((r < 0.0) && (gamma != 0.0)) stpc = *stp + r * (*stx - *stp);
        else if (*stp > *stx) stpc = stmax;
        else stpc = stmin;
        stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
        if (*brackt) {
            if (fabs(*stp - stpc) < fabs(*stp - stpq)) stpf = stpc;
            else stpf = stpq;
        } else {
            if (fabs(*stp - stpc) > fabs(*stp - stpq)) stpf = stpc;
            else stpf = stpq;
        }
    } else {
        /* case 4: lower value, same sign, derivative magnitude does not decrease */
        info = 4;
        bound = 0;
        if (*brackt) {
            theta = 3 * (fp - *fy) / (*sty - *stp) + *dy + dp;
            s = absmax3(theta, *dy, dp);
            gamma = s * sqrt(((theta / s) * (theta / s)) - (*dy / s) * (dp / s));
            if (*stp > *sty) gamma = -gamma;
            p = (gamma - dp) + theta;
            q = ((gamma - dp) + gamma) + *dy;
            r = p / q;
            stpc = *stp + r * (*sty - *stp);
            stpf = stpc;
        } else if (*stp > *stx)
            stpf = stmax;
        else
            stpf = stmin;
    }

    /* update the interval of uncertainty */
    if (fp > *fx) {
        *sty = *stp; *fy = fp; *dy = dp;
    } else {
        if (sgnd < 0.0) { *sty = *stx; *fy = *fx; *dy = *dx; }
        *stx = *stp; *fx = fp; *dx = dp;
    }

    /* new step, safeguarded */
    stpf = dmin(stmax, stpf);
    stpf = dmax(stmin, stpf);
    *stp = stpf;
    if (*brackt && bound) {
        if (*sty > *stx) *stp = dmin(*stx + 0.66 * (*sty - *stx), *stp);
        else *stp = dmax(*stx + 0.66 * (*sty - *stx), *stp);
    }
    return info;
}

/* lineSearchMTFusion body after the dginit sign handling: [fusion.h]:485-791.
 * Constants [fusion.h]:400-408. */
double oracle_mt_linesearch(oracle_phi_fn phi, void *ctx, double finit, double dginit, int *nfev_out,
                            int *info_out)
{
    double stp = 1.0;
    const double recoverystep = 0.1;
    const double ftol = 0.11111, gtol = 0.99999;
    const double stpmax = 4.0, stpmin = 0.001;
    const int maxfev = 40;
    const double xtol = 0.01;

    int info = 0, infoc = 1;
    int brackt = 0, stage1 = 1, nfev = 0;
    double dgtest = ftol * dginit;
    double width = stpmax - stpmin;
    double width1 = 2 * width;
    double stx = 0.0, fx = finit, dgx = dginit;
    double sty = 0.0, fy = finit, dgy = dginit;
    double stmin, stmax;
    double fm, fxm, fym, dgm, dgxm, dgym;

    for (;;) {
        if (brackt) {
            stmin = dmin(stx, sty);
            stmax = dmax(stx, sty);
        } else {
            stmin = stx;
            stmax = stp + 4 * (stp - stx);
        }
        stp = dmax(stp, stpmin);
        stp = dmin(stp, stpmax);

        if ((brackt && ((stp <= stmin) || (stp >= stmax))) || (nfev >= maxfev - 1) || (infoc == 0) ||
            (brackt && (stmax - stmin <= xtol * stmax)))
            stp = stx;

        double dg = 0.0;
        double f = phi(ctx, stp, &dg);
        nfev++;

        double ftest1 = finit + stp * dgtest;

        if ((brackt && ((stp <= stmin) || (stp >= stmax))) || (infoc == 0)) info = 6;
        if ((stp == stpmax) && (f <= ftest1) && (dg <= dgtest)) info = 5;
        if ((stp == stpmin) && ((f > ftest1) || (dg >= dgtest))) info = 4;
        if (nfev >= maxfev) info = 3;
        if (brackt && (stmax - stmin <= xtol * stmax)) info = 2;
        if ((f <= ftest1) && (fabs(dg) <= gtol * (-dginit))) info = 1;

        if (info != 0) {
            if (info != 1) stp = recoverystep;
            if (nfev_out) *nfev_out = nfev;
            if (info_out) *info_out = info;
            return stp;
        }

        if (stage1 && (f <= ftest1) && (dg >= dmin(ftol, gtol) * dginit)) stage1 = 0;

        if (stage1 && (f <= fx) && (f > ftest1)) {
            fm = f - stp * dgtest;
            fxm = fx - stx * dgtest;
            fym = fy - sty * dgtest;
            dgm = dg - dgtest;
            dgxm = dgx - dgtest;
            dgym = dgy - dgtest;
            infoc = oracle_mt_cstep(&stx, &fxm, &dgxm, &sty, &fym, &dgym, &stp, fm, dgm, &brackt, stmin, stmax);
            fx = fxm + stx * dgtest;
            fy = fym + sty * dgtest;
            dgx = dgxm + dgtest;
            dgy = dgym + dgtest;
        } else {
            infoc = oracle_mt_cstep(&stx, &fx, &dgx, &sty, &fy, &dgy, &stp, f, dg, &brackt, stmin, stmax);
        }

        if (brackt) {
            if (fabs(sty - stx) >= 0.66 * width1) stp = stx + 0.5 * (sty - stx);
            width1 = width;
            width = fabs(sty - stx);
        }
    }
}

/* ------------------------------------------------------------------ */
/* NDTMatcherD2D::match                                                 */
/* ------------------------------------------------------------------ */

typedef struct {
    const oracle_map *target;
    const ocell *cells; /* nextNDT */
    ocell *scratch;     /* sourceNDTHere */
    size_t n;
    const double *incr; /* full 6-vector */
    const oracle_match_params *prm;
    int *fevals;
} ls_ctx;

/* one trial of the line search: [fusion.h]:556-639 */
static double ls_phi(void *vctx, double stp, double *dg)
{
    ls_ctx *c = (ls_ctx *)vctx;
    double pincr[6], ps[16], g[6], H[36];
    for (int a = 0; a < 6; a++) pincr[a] = stp * c->incr[a];
    oracle_pose_to_T(pincr, ps);
    transform_cells(c->cells, c->scratch, c->n, ps);
    double f = derivatives_cells(c->target, c->scratch, c->n, c->prm->n_neighbours, 0, c->prm->lfd1,
                                 c->prm->lfd2, g, H);
    (*c->fevals)++;
    double d = 0;
    for (int a = 0; a < 6; a++) d += c->incr[a] * g[a];
    *dg = d;
    return f;
}

/* NDTMatcherD2D::lineSearchMT == lineSearchMTFusion minus the feature terms:
 * [fusion.h]:439-483 (initial evaluation, direction flip, recovery step). */
static double line_search_mt(double incr[6], const oracle_map *target, const ocell *cells, ocell *scratch,
                             size_t n, const oracle_match_params *prm, int *fevals)
{
    double g[6], H[36];
    double score_init =
        derivatives_cells(target, cells, n, prm->n_neighbours, 0, prm->lfd1, prm->lfd2, g, H);
    (*fevals)++;
    double dginit = 0;
    for (int a = 0; a < 6; a++) dginit += incr[a] * g[a];
    if (dginit >= 0.0) {
        for (int a = 0; a < 6; a++) incr[a] = -incr[a];
        dginit = -dginit;
        if (dginit >= 0.0) return 0.1; /* recoverystep */
    }
    ls_ctx ctx = {target, cells, scratch, n, incr, prm, fevals};
    return oracle_mt_linesearch(ls_phi, &ctx, score_init, dginit, NULL, NULL);
}

/* Gauss-Jordan with partial pivoting: Eigen's Tcov.inverse() ([fusion.h]:845) */
static int invert6(const double *A, double *inv)
{
    double a[6][12];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) { a[i][j] = A[i * 6 + j]; a[i][6 + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 6; c++) {
        int piv = c;
        for (int r = c + 1; r < 6; r++)
            if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (a[piv][c] == 0.0) return 0;
        if (piv != c)
            for (int j = 0; j < 12; j++) { double t = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = t; }
        double d = a[c][c];
        for (int j = 0; j < 12; j++) a[c][j] /= d;
        for (int r = 0; r < 6; r++) {
            if (r == c) continue;
            double f = a[r][c];
            if (f != 0.0)
                for (int j = 0; j < 12; j++) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) inv[i * 6 + j] = a[i][6 + j];
    return 1;
}

static int match_common(const oracle_map *target, const oracle_map *source, double T[16],
                        const oracle_match_params *prm, const double *Q /* Tcov^-1 or NULL */, oracle_match_result *res);

int oracle_match_d2d(const oracle_map *target, const oracle_map *source, double T[16],
                     const oracle_match_params *prm, oracle_match_result *res)
{
    return match_common(target, source, T, prm, NULL, res);
}

/* ndt_feature::matchFusion ([fusion.h]:797-1155) with useNDT = true, useFeat = false (empty feature
 * maps: their derivativesNDT terms are zero), useTikhonovRegularization = false.  With
 * useSoftConstraints the score / gradient / Hessian get the Mahalanobis terms of X = pose_local_v
 * ([fusion.h]:875-890, 1098-1110); the step length comes from NDTMatcherD2D::lineSearchMT because the
 * result of lineSearchMTFusionTcov is overwritten ([fusion.h]:1008-1023: step_size_feat == 0 ->
 * step_size = max(step_size_ndt, 0)); that discarded search can only flip the increment when
 * dginit >= 0, which the loop has just excluded ([fusion.h]:976), so it is not evaluated here. */
int oracle_match_fusion(const oracle_map *target, const oracle_map *source, double T[16],
                        const oracle_match_params *prm, const double Tcov[36], int use_soft_constraints,
                        oracle_match_result *res)
{
    double Q[36];
    if (!use_soft_constraints) return match_common(target, source, T, prm, NULL, res);
    if (!invert6(Tcov, Q)) return -2;
    return match_common(target, source, T, prm, Q, res);
}

static int match_common(const oracle_map *target, const oracle_map *source, double T[16],
                        const oracle_match_params *prm, const double *Q, oracle_match_result *res)
{
    int dofs[6], nd = 0;
    for (int a = 0; a < 6; a++)
        if (prm->dof_mask & (1 << a)) dofs[nd++] = a;
    if (nd == 0) return -1;

    int convergence = 0, ret = 1, itr_ctr = 0, fevals = 0, exit_code = 0;
    double score_best = DBL_MAX, score_here = 0;
    double Tbest[16];
    if (!prm->use_initial_guess) {
        double z[6] = {0, 0, 0, 0, 0, 0};
        oracle_pose_to_T(z, T);
    }
    memcpy(Tbest, T, sizeof Tbest);

    size_t n;
    double pose_local[6] = {0, 0, 0, 0, 0, 0};
    ocell *next = pseudo_transform(source, T, &n); /* [fusion.h]:840 */
    ocell *scratch = (ocell *)calloc(n ? n : 1, sizeof(ocell));
    double g6[6], H36[36];

    while (!convergence) {
        score_here = derivatives_cells(target, next, n, prm->n_neighbours, 1, prm->lfd1, prm->lfd2, g6, H36);
        fevals++;
        if (Q) { /* [fusion.h]:875-890 */
            double gq[6], Hq[36];
            score_here += oracle_mahalanobis(pose_local, Q, gq, Hq);
            for (int i = 0; i < 6; i++) g6[i] += gq[i];
            for (int i = 0; i < 36; i++) H36[i] += Hq[i];
        }
        /* restrict to the active dofs (6-DoF: identity) */
        double g[6], H[36];
        for (int i = 0; i < nd; i++) {
            g[i] = g6[dofs[i]];
            for (int j = 0; j < nd; j++) H[i * nd + j] = H36[dofs[i] * 6 + dofs[j]];
        }
        if (score_here < score_best) { /* [fusion.h]:914-920 */
            memcpy(Tbest, T, sizeof Tbest);
            score_best = score_here;
        }
        double gnorm = 0;
        for (int i = 0; i < nd; i++) gnorm += g[i] * g[i];
        gnorm = sqrt(gnorm);

        /* [fusion.h]:922-940 */
        double ev[6], V[36];
        oracle_eig_sym(nd, H, ev, V);
        double minC = ev[0], maxC = ev[nd - 1];
        if (minC < 0) {
            double regularizer = gnorm;
            regularizer = (regularizer + minC > 0) ? regularizer : 0.001 * maxC - minC;
            for (int i = 0; i < nd; i++) ev[i] += regularizer;
            for (int i = 0; i < nd; i++)
                for (int j = 0; j < nd; j++) {
                    double s = 0;
                    for (int k = 0; k < nd; k++) s += V[i * nd + k] * ev[k] * V[j * nd + k];
                    H[i * nd + j] = s;
                }
        }
        /* [fusion.h]:943-965 */
        if (gnorm <= prm->delta_score) {
            if (score_here > score_best) memcpy(T, Tbest, sizeof Tbest);
            exit_code = 1;
            goto done_early;
        }
        /* [fusion.h]:966-997 */
        double dx[6], incr[6] = {0, 0, 0, 0, 0, 0};
        oracle_ldlt_solve(nd, H, g, dx);
        double dginit = 0;
        for (int i = 0; i < nd; i++) {
            incr[dofs[i]] = -dx[i];
            dginit += -dx[i] * g[i];
        }
        if (dginit > 0) {
            if (score_here > score_best) memcpy(T, Tbest, sizeof Tbest);
            exit_code = 2;
            goto done_early;
        }
        /* [fusion.h]:1000-1032 */
        double step_size = 1.0;
        if (prm->step_control) {
            /* the line search sees only the active dofs' gradient through incr (inactive entries are 0) */
            step_size = line_search_mt(incr, target, next, scratch, n, prm, &fevals);
        }
        double inorm = 0;
        for (int a = 0; a < 6; a++) {
            incr[a] *= step_size;
            inorm += incr[a] * incr[a];
        }
        inorm = sqrt(inorm);
        for (int a = 0; a < 6; a++) pose_local[a] += incr[a]; /* [fusion.h]:1045 */
        /* [fusion.h]:1035-1066 */
        double TR[16];
        oracle_pose_to_T(incr, TR);
        T_mul(TR, T, T);
        transform_cells(next, next, n, TR);
        /* [fusion.h]:1070-1080 */
        if (itr_ctr > 0) convergence = (inorm < prm->delta_score);
        if (itr_ctr > prm->itr_max) {
            convergence = 1;
            ret = 0;
            exit_code = 3;
        }
        itr_ctr++;
    }
    /* [fusion.h]:1085-1121 */
    score_here = derivatives_cells(target, next, n, prm->n_neighbours, 0, prm->lfd1, prm->lfd2, g6, H36);
    fevals++;
    if (Q) { /* [fusion.h]:1098-1110 */
        double gq[6], Hq[36];
        score_here += oracle_mahalanobis(pose_local, Q, gq, Hq);
    }
    if (score_here > score_best) memcpy(T, Tbest, sizeof Tbest);

done_early:
    if (res) {
        res->converged = ret;
        res->iterations = itr_ctr;
        res->fevals = fevals;
        res->score = (score_here > score_best) ? score_best : score_here;
        res->exit_code = exit_code;
    }
    free(next);
    free(scratch);
    return 0;
}
