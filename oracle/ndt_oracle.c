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
#include <stdio.h>

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
    int32_t *cell_of_slot; /* -1 = no points / no Gaussian ever (the NDTCell may still exist: see occ) */
    ocell *cells;
    size_t ncells, capcells;
    /* NDTCell::occ of every slot (float, like upstream).  NDTMap::initialize allocates every cell with occ = 0;
     * a lazily allocated map has occ = 0 in the cells that do not exist, which nobody can tell apart (every reader
     * skips occ == 0: ndt_feature_node.h:224, 237). */
    float *occ;
    /* points waiting for computeNDTCells (NDTCell::points_ of the cells in update_set), in insertion order */
    float *pend_xyz;
    int32_t *pend_cell;
    size_t npend, cappend;
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
    m->occ = (float *)calloc(m->nslots, sizeof(float));
    if (!m->occ) { free(m->cell_of_slot); free(m); return NULL; }
    return m;
}

static void map_clear(oracle_map *m)
{
    for (size_t i = 0; i < m->nslots; i++) { m->cell_of_slot[i] = -1; m->occ[i] = 0.0f; }
    free(m->cells); m->cells = NULL; m->ncells = m->capcells = 0;
    free(m->pend_xyz); m->pend_xyz = NULL;
    free(m->pend_cell); m->pend_cell = NULL;
    m->npend = m->cappend = 0;
}

void oracle_map_destroy(oracle_map *m)
{
    if (!m) return;
    map_clear(m);
    free(m->cell_of_slot);
    free(m->occ);
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

/* LazyGrid::addPoint: the point joins points_ of its cell (created on demand) and the cell joins update_set */
static int pend_point(oracle_map *m, const float *q)
{
    double p[3] = {q[0], q[1], q[2]};
    int idx[3];
    index_for_point(m, p, idx);
    if (!idx_inside(m, idx)) return 0;
    int c = cell_for_slot_create(m, slot_of(m, idx), idx);
    if (c < 0) return -1;
    if (m->npend == m->cappend) {
        size_t nc = m->cappend ? 2 * m->cappend : 4096;
        float *px = (float *)realloc(m->pend_xyz, nc * 3 * sizeof(float));
        if (!px) return -1;
        m->pend_xyz = px;
        int32_t *pc = (int32_t *)realloc(m->pend_cell, nc * sizeof(int32_t));
        if (!pc) return -1;
        m->pend_cell = pc;
        m->cappend = nc;
    }
    m->pend_xyz[3 * m->npend] = q[0]; m->pend_xyz[3 * m->npend + 1] = q[1]; m->pend_xyz[3 * m->npend + 2] = q[2];
    m->pend_cell[m->npend++] = c;
    return 0;
}

int oracle_map_load_points(oracle_map *m, const float *xyz, size_t n, size_t stride, double range_limit,
                           const double *range_origin)
{
    map_clear(m);
    /* NDTMap::loadPointCloud: skip NaN, skip ||p|| > range_limit, LazyGrid::addPoint drops
     * points whose index falls outside the grid. */
    for (size_t i = 0; i < n; i++) {
        const float *q = xyz + i * stride;
        if (isnan(q[0]) || isnan(q[1]) || isnan(q[2])) continue;
        if (range_limit > 0) {
            double d[3] = {q[0], q[1], q[2]};
            if (range_origin)
                for (int a = 0; a < 3; a++) d[a] -= range_origin[a];
            if (sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) > range_limit) continue;
        }
        if (pend_point(m, q) < 0) return -1;
    }
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

/* NDTCell::updateOccupancy(occ_val, max_occu): float arithmetic, clamped to +-max_occu */
static void update_occupancy(oracle_map *m, size_t slot, float occ_val, float max_occu)
{
    float o = m->occ[slot] + occ_val;
    if (o > max_occu) o = max_occu;
    if (o < -max_occu) o = -max_occu;
    m->occ[slot] = o;
}

/* NDTMap::computeNDTCells(mode = SAMPLE_VARIANCE, maxnumpoints, occupancy_limit, origin, sensor_noise)
 * (fuser_hmt.cpp:94, 486; :227 and ndt_odom_debug.cpp:179 with the defaults 1e9 / 255) over update_set ->
 * NDTCell::computeGaussian (perception_oru; origin and sensor_noise are unused on this branch):
 *   occupancy += n * log(0.6 / 0.4), clamped;  occ <= 0 -> hasGaussian_ = false, points dropped;
 *   no Gaussian yet and n < n_min -> points dropped;
 *   no Gaussian yet: mean, cov = sum (p - mean)(p - mean)^T / (n - 1), N = n;
 *   Gaussian already: Chan's pairwise update of (N, N mean, (N - 1) cov) with the new batch, N saturating at
 *   maxnumpoints ("sliding average");
 *   rescaleCovariance. */
int oracle_map_compute_cells_full(oracle_map *m, int n_min, double eval_factor, double maxnumpoints, double occupancy_limit)
{
    if (!m->npend) return 0;
    /* group the pending points by cell, insertion order kept */
    size_t *begin = (size_t *)calloc(m->ncells + 1, sizeof(size_t));
    size_t *fill = (size_t *)calloc(m->ncells ? m->ncells : 1, sizeof(size_t));
    float *pts = (float *)malloc(m->npend * 3 * sizeof(float));
    if (!begin || !fill || !pts) { free(begin); free(fill); free(pts); return -1; }
    for (size_t i = 0; i < m->npend; i++) begin[m->pend_cell[i] + 1]++;
    for (size_t c = 0; c < m->ncells; c++) begin[c + 1] += begin[c];
    for (size_t i = 0; i < m->npend; i++) {
        int c = m->pend_cell[i];
        size_t o = (begin[c] + fill[c]++) * 3;
        pts[o] = m->pend_xyz[3 * i]; pts[o + 1] = m->pend_xyz[3 * i + 1]; pts[o + 2] = m->pend_xyz[3 * i + 2];
    }
    for (size_t c = 0; c < m->ncells; c++) {
        ocell *ce = &m->cells[c];
        size_t b = begin[c], e = begin[c + 1], n = e - b;
        if (n == 0) continue;                                   /* not in update_set */
        size_t slot = slot_of(m, ce->idx);
        double logoddlikoccu = (double)n * log(0.6 / (1.0 - 0.6));
        update_occupancy(m, slot, (float)logoddlikoccu, (float)occupancy_limit);   /* > 0.4 for every n >= 1 */
        if (m->occ[slot] <= 0) { ce->has_gaussian = 0; continue; }
        if (!ce->has_gaussian && (int)n < n_min) continue;
        if (!ce->has_gaussian) {
            double mean[3] = {0, 0, 0};
            for (size_t i = b; i < e; i++)
                for (int a = 0; a < 3; a++) mean[a] += (double)pts[i * 3 + a];
            for (int a = 0; a < 3; a++) mean[a] /= (double)n;
            mat3 S = m3_zero();
            for (size_t i = b; i < e; i++) {
                double d[3];
                for (int a = 0; a < 3; a++) d[a] = (double)pts[i * 3 + a] - mean[a];
                for (int r = 0; r < 3; r++)
                    for (int q = 0; q < 3; q++) S.m[r][q] += d[r] * d[q];
            }
            for (int r = 0; r < 3; r++)
                for (int q = 0; q < 3; q++) ce->cov.m[r][q] = S.m[r][q] / (double)(n - 1);
            for (int a = 0; a < 3; a++) ce->mean[a] = mean[a];
            ce->n = (int)n;
            rescale_covariance(ce, eval_factor);
        } else {
            double N = (double)ce->n, dn = (double)n;
            double meanSum[3], T2[3] = {0, 0, 0}, m2[3];
            mat3 covSum, c2 = m3_zero();
            for (int a = 0; a < 3; a++) meanSum[a] = ce->mean[a] * N;
            for (int r = 0; r < 3; r++)
                for (int q = 0; q < 3; q++) covSum.m[r][q] = ce->cov.m[r][q] * (N - 1.0);
            for (size_t i = b; i < e; i++)
                for (int a = 0; a < 3; a++) T2[a] += (double)pts[i * 3 + a];
            for (int a = 0; a < 3; a++) m2[a] = T2[a] / dn;
            for (size_t i = b; i < e; i++) {
                double d[3];
                for (int a = 0; a < 3; a++) d[a] = (double)pts[i * 3 + a] - m2[a];
                for (int r = 0; r < 3; r++)
                    for (int q = 0; q < 3; q++) c2.m[r][q] += d[r] * d[q];
            }
            double w1 = N / (dn * (N + dn)), w2 = dn / N, c3[3];
            for (int a = 0; a < 3; a++) c3[a] = meanSum[a] * w2 - T2[a];
            for (int r = 0; r < 3; r++)
                for (int q = 0; q < 3; q++) covSum.m[r][q] += c2.m[r][q] + w1 * (c3[r] * c3[q]);
            for (int a = 0; a < 3; a++) meanSum[a] += T2[a];
            N += dn;
            if (maxnumpoints > 0 && maxnumpoints < N) {
                for (int a = 0; a < 3; a++) meanSum[a] *= maxnumpoints / N;
                for (int r = 0; r < 3; r++)
                    for (int q = 0; q < 3; q++) covSum.m[r][q] *= (maxnumpoints - 1.0) / (N - 1.0);
                N = maxnumpoints;
            }
            for (int a = 0; a < 3; a++) ce->mean[a] = meanSum[a] / N;
            for (int r = 0; r < 3; r++)
                for (int q = 0; q < 3; q++) ce->cov.m[r][q] = covSum.m[r][q] / (N - 1.0);
            ce->n = (int)N;
            rescale_covariance(ce, eval_factor);
        }
    }
    free(begin); free(fill); free(pts);
    m->npend = 0;
    return 0;
}

/* NDTMap::computeNDTCells(CELL_UPDATE_MODE_SAMPLE_VARIANCE) with its default arguments */
int oracle_map_compute_cells(oracle_map *m, int n_min, double eval_factor)
{
    return oracle_map_compute_cells_full(m, n_min, eval_factor, 1e9, 255.0);
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
/* NDTMap::addPointCloud (ray-traced insert) and the occupancy readers    */
/* ------------------------------------------------------------------ */

/* 3x3 inverse by cofactors (NDTCell::icov_; upstream builds it from the eigen-decomposition of
 * rescaleCovariance: the same matrix up to rounding) */
static int m3_inverse_plain(mat3 a, mat3 *inv)
{
    mat3 t;
    double c00 = a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1];
    double c01 = a.m[1][2] * a.m[2][0] - a.m[1][0] * a.m[2][2];
    double c02 = a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0];
    double det = a.m[0][0] * c00 + a.m[0][1] * c01 + a.m[0][2] * c02;
    if (det == 0.0 || det != det) return 0;
    double id = 1.0 / det;
    t.m[0][0] = c00 * id; t.m[1][0] = c01 * id; t.m[2][0] = c02 * id;
    t.m[0][1] = (a.m[0][2] * a.m[2][1] - a.m[0][1] * a.m[2][2]) * id;
    t.m[1][1] = (a.m[0][0] * a.m[2][2] - a.m[0][2] * a.m[2][0]) * id;
    t.m[2][1] = (a.m[0][1] * a.m[2][0] - a.m[0][0] * a.m[2][1]) * id;
    t.m[0][2] = (a.m[0][1] * a.m[1][2] - a.m[0][2] * a.m[1][1]) * id;
    t.m[1][2] = (a.m[0][2] * a.m[1][0] - a.m[0][0] * a.m[1][2]) * id;
    t.m[2][2] = (a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0]) * id;
    *inv = t;
    return 1;
}

/* The emptiness evidence ONE beam leaves in ONE traversed cell that holds a Gaussian
 * (NDTMap::addPointCloud + NDTCell::computeMaximumLikelihoodAlongLine + getLikelihood, perception_oru; SURVEY
 * App. A): X = the point of maximum likelihood on the beam, lik = exp(-(X - mean)^T icov (X - mean) / 2); a maximum
 * behind the measured end is ignored; the closer X is to the end point the less the beam says about the cell
 * being empty (distance-dependent sensor noise).  origin: the sensor (double, and its float copy `po` as the
 * pcl::PointXYZ upstream builds); end point pe (float).  Returns 0 when the beam leaves the cell untouched, else
 * writes the (negative) log-odds update as the float NDTCell::updateOccupancy receives. */
int oracle_beam_evidence(const double mean[3], const double cov9[9], const double origin[3], const float pe[3],
                         double sensor_noise, float *logodd)
{
    mat3 C, ic;
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) C.m[a][b] = cov9[3 * a + b];
    if (!m3_inverse_plain(C, &ic)) return 0;
    float po[3] = {(float)origin[0], (float)origin[1], (float)origin[2]};
    double v1[3] = {po[0], po[1], po[2]}, v2[3] = {pe[0], pe[1], pe[2]};
    double d[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
    double nl = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    vec3 L = {{d[0] / nl, d[1] / nl, d[2] / nl}};
    vec3 A = m3_v(ic, L);
    vec3 B = {{v2[0] - mean[0], v2[1] - mean[1], v2[2] - mean[2]}};
    double sigma = A.v[0] * L.v[0] + A.v[1] * L.v[1] + A.v[2] * L.v[2];
    if (sigma == 0) return 0; /* upstream returns 1.0 without setting `out` and then reads it: unreachable for a regular Gaussian */
    double t = -(A.v[0] * B.v[0] + A.v[1] * B.v[1] + A.v[2] * B.v[2]) / sigma;
    double X[3], lik;
    for (int a = 0; a < 3; a++) X[a] = L.v[a] * t + v2[a];
    {   /* getLikelihood(pcl::PointXYZ): the point is rounded to float first */
        float xf[3] = {(float)X[0], (float)X[1], (float)X[2]};
        vec3 w = {{(double)xf[0] - mean[0], (double)xf[1] - mean[1], (double)xf[2] - mean[2]}};
        double q = v3_dot(w, m3_v(ic, w));
        lik = (q != q) ? -1.0 : exp(-q / 2);
    }
    /* l = |end - origin| and dist = |origin - X| use the DOUBLE origin */
    double e[3] = {v2[0] - origin[0], v2[1] - origin[1], v2[2] - origin[2]};
    double l = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    double o[3] = {origin[0] - X[0], origin[1] - X[1], origin[2] - X[2]};
    double dist = sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
    if (dist > l) return 0;
    double g[3] = {X[0] - v2[0], X[1] - v2[1], X[2] - v2[2]};
    double l2target = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    double sigma_dist = 0.5 * (dist / 30.0);
    double snoise = sigma_dist + sensor_noise;
    double thr = exp(-0.5 * (l2target * l2target) / (snoise * snoise));
    lik *= (1.0 - thr);
    if (lik < 0.3) return 0;
    lik = 0.1 * lik + 0.5;
    *logodd = (float)log((1.0 - lik) / lik);
    return 1;
}

/* LazyGrid::traceLine: the cells a beam crosses, SAMPLED every `res` along the beam (N = (int)(l / res) steps,
 * samples i = 1 .. N - 2: the walk stops about two cells short of the end point), sample coordinates rounded to
 * float (pcl::PointXYZ), consecutive samples in the same cell visited once.  visit(slot) for every in-grid cell. */
typedef void (*visit_fn)(void *ctx, size_t slot);
static void trace_line(const oracle_map *m, const double origin[3], const float pe[3], visit_fn visit, void *ctx)
{
    double diff[3] = {(double)pe[0] - origin[0], (double)pe[1] - origin[1], (double)pe[2] - origin[2]};
    double l = sqrt(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]);
    int N = (int)(l / m->res);
    if (N <= 2) return;
    double step[3] = {diff[0] / (float)N, diff[1] / (float)N, diff[2] / (float)N};
    int io[3] = {0, 0, 0};   /* idxo = idyo = idzo = 0: a first sample in cell (0,0,0) is skipped like upstream */
    for (int i = 0; i < N - 2; i++) {
        float pt[3];
        for (int a = 0; a < 3; a++) pt[a] = (float)(origin[a] + ((float)(i + 1)) * step[a]);
        double p[3] = {pt[0], pt[1], pt[2]};
        int idx[3];
        index_for_point(m, p, idx);
        if (idx[0] == io[0] && idx[1] == io[1] && idx[2] == io[2]) continue;
        io[0] = idx[0]; io[1] = idx[1]; io[2] = idx[2];
        if (!idx_inside(m, idx)) continue;
        visit(ctx, slot_of(m, idx));
    }
}

typedef struct {
    oracle_map *m;
    const double *origin;
    const float *pe;
    double sensor_noise;
    float occupancy_limit;
    int order_free;
    const unsigned char *had_gaussian; /* order_free: hasGaussian_ of every cell when the call started */
    long long *delta;                  /* order_free: exact sum of the float updates, in units of 2^-32 */
} beam_ctx;

static void visit_cell(void *vctx, size_t slot)
{
    beam_ctx *b = (beam_ctx *)vctx;
    oracle_map *m = b->m;
    int32_t c = m->cell_of_slot[slot];
    int gauss = b->order_free ? (int)b->had_gaussian[slot] : (c >= 0 && m->cells[c].has_gaussian);
    float upd;
    if (gauss) {
        const ocell *ce = &m->cells[c];
        double cov9[9];
        for (int a = 0; a < 3; a++)
            for (int q = 0; q < 3; q++) cov9[3 * a + q] = ce->cov.m[a][q];
        if (!oracle_beam_evidence(ce->mean, cov9, b->origin, b->pe, b->sensor_noise, &upd)) return;
    } else {
        upd = -0.2f;   /* seen empty, no Gaussian to argue with */
    }
    if (b->order_free) {
        b->delta[slot] += llrint((double)upd * 4294967296.0);   /* exact: a float below 1 in magnitude */
        return;
    }
    update_occupancy(m, slot, upd, b->occupancy_limit);
    if (m->occ[slot] <= 0 && c >= 0) m->cells[c].has_gaussian = 0;
}

/* NDTMap::addPointCloud(origin, pc, classifierTh, maxz, sensor_noise, occupancy_limit)
 * (fuser_hmt.cpp:92: (Tnow_sensor, cloud, 0.1, 100.0, 0.1); :485: (spose, cloud_orig, 0.06, 25)) on a map that went
 * through NDTMap::initialize (isFirstLoad_ == false, every cell allocated).  classifierTh is unused upstream.
 * order_free == 0: the reference's sequential semantics (beam after beam; a cell that loses its Gaussian half way
 *                  through the cloud is treated as empty by the remaining beams; float accumulation);
 * order_free == 1: what the HIP path implements (DESIGN.md "Deviations"): every beam sees the cells as they were
 *                  when the call started, the updates of a cell are summed exactly and applied once. */
int oracle_map_add_point_cloud(oracle_map *m, const double origin[3], const float *xyz, size_t n, size_t stride,
                               double maxz, double sensor_noise, double occupancy_limit, int order_free)
{
    const double max_range = 200.0;
    beam_ctx b = {m, origin, NULL, sensor_noise, (float)occupancy_limit, order_free, NULL, NULL};
    unsigned char *had = NULL;
    long long *delta = NULL;
    if (order_free) {
        had = (unsigned char *)calloc(m->nslots, 1);
        delta = (long long *)calloc(m->nslots, sizeof(long long));
        if (!had || !delta) { free(had); free(delta); return -1; }
        for (size_t s = 0; s < m->nslots; s++) {
            int32_t c = m->cell_of_slot[s];
            had[s] = (c >= 0 && m->cells[c].has_gaussian) ? 1 : 0;
        }
        b.had_gaussian = had;
        b.delta = delta;
    }
    for (size_t i = 0; i < n; i++) {
        const float *q = xyz + i * stride;
        if (isnan(q[0]) || isnan(q[1]) || isnan(q[2])) continue;
        double d[3] = {(double)q[0] - origin[0], (double)q[1] - origin[1], (double)q[2] - origin[2]};
        if (sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) > max_range) continue;
        if ((double)q[2] > maxz) continue;                    /* traceLine returns false: the point is dropped */
        b.pe = q;
        trace_line(m, origin, q, visit_cell, &b);
        if (pend_point(m, q) < 0) { free(had); free(delta); return -1; }
    }
    if (order_free) {
        for (size_t s = 0; s < m->nslots; s++) {
            if (!delta[s]) continue;
            float o = (float)((double)m->occ[s] + (double)delta[s] * (1.0 / 4294967296.0));
            if (o > (float)occupancy_limit) o = (float)occupancy_limit;
            if (o < -(float)occupancy_limit) o = -(float)occupancy_limit;
            m->occ[s] = o;
            int32_t c = m->cell_of_slot[s];
            if (o <= 0 && c >= 0) m->cells[c].has_gaussian = 0;
        }
        free(had);
        free(delta);
    }
    return 0;
}

void oracle_map_size(const oracle_map *m, int size[3])
{
    for (int a = 0; a < 3; a++) size[a] = m->size[a];
}

void oracle_map_occupancy(const oracle_map *m, float *occ_out)
{
    memcpy(occ_out, m->occ, m->nslots * sizeof(float));
}

/* NDTCell::getOccupancyRescaled: 1 - 1 / (1 + exp(occ)) in FLOAT arithmetic.  expf is restated as the correctly
 * rounded value float(exp(double)) (glibc's expf is correctly rounded for all but a handful of inputs), so that the
 * HIP path can reproduce it bit for bit. */
float oracle_occupancy_rescaled(float occ)
{
    float e = (float)exp((double)occ);
    float o = 1.0f - 1.0f / (1.0f + e);
    return o > 1 ? 1 : (o < 0 ? 0 : o);
}

/* ndt_feature::overlapNDTOccupancyScore(ref, mov, T)  (ndt_feature_node.h:213-252; used at
 * ndt_feature_graph.cpp:175, 338-340): over the cells of `mov` that carry a reading, the squared difference of the
 * rescaled occupancies with the `ref` cell their transformed centre falls into. */
double oracle_overlap_score(const oracle_map *ref, const oracle_map *mov, const double T[16], long long *nb_sum_out)
{
    double diff_sum = 0;
    long long nb_sum = 0;
    for (size_t s = 0; s < mov->nslots; s++) {
        double mov_occ = oracle_occupancy_rescaled(mov->occ[s]);
        if (mov_occ == 0.5) continue;
        size_t iz = s % (size_t)mov->size[2], iy = (s / (size_t)mov->size[2]) % (size_t)mov->size[1],
               ix = s / ((size_t)mov->size[2] * (size_t)mov->size[1]);
        /* NDTCell::getCenter(): pcl::PointXYZ (float) of the cell centre */
        float cf[3] = {(float)(mov->centre[0] + ((double)ix - mov->size[0] / 2) * mov->res),
                       (float)(mov->centre[1] + ((double)iy - mov->size[1] / 2) * mov->res),
                       (float)(mov->centre[2] + ((double)iz - mov->size[2] / 2) * mov->res)};
        double e[3] = {cf[0], cf[1], cf[2]}, t[3];
        for (int r = 0; r < 3; r++) t[r] = T[0 * 4 + r] * e[0] + T[1 * 4 + r] * e[1] + T[2 * 4 + r] * e[2] + T[12 + r];
        float pf[3] = {(float)t[0], (float)t[1], (float)t[2]};
        double p[3] = {pf[0], pf[1], pf[2]};
        int idx[3];
        index_for_point(ref, p, idx);
        if (!idx_inside(ref, idx)) continue;
        double ref_occ = oracle_occupancy_rescaled(ref->occ[slot_of(ref, idx)]);
        if (ref_occ != 0.5) {
            nb_sum++;
            double diff = mov_occ - ref_occ;
            diff_sum += diff * diff;
        }
    }
    if (nb_sum_out) *nb_sum_out = nb_sum;
    if (nb_sum == 0) return 1.;
    return diff_sum / (1. * nb_sum);
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

/* the terms of ONE source cell added to score / g / H (the body of derivativesNDT's loop over the source cells) */
static void derivatives_one_cell(const oracle_map *target, const ocell *sc, int n_neighbours, int with_hessian, double lfd1,
                                 double lfd2, double *score, double g[6], double H[36])
{
    vec3 mm = {{sc->mean[0], sc->mean[1], sc->mean[2]}};
    mat3 CM = sc->cov;
    local_derivs L;
    compute_derivatives_local(mm, CM, with_hessian, &L);
    int ic[3];
    index_for_point(target, sc->mean, ic);
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
                *score += sh;
            }
}

/* TEST KNOB (this file is test infrastructure): the ORDER in which the source cells' terms are added.  0 = the
 * reference's order (source cells one after the other).  1 = the same cells in reverse; 2 = dealt to 8 shares (cell i ->
 * share i mod 8), a partial sum per share, the partials added in share order -- the shape of the HIP matcher's sums;
 * 3 = 8 shares, reversed inside a share; 4..15 = the reference order with every sum moved by <= 8 ulp afterwards; 16 and up = the reference's sums,
 * the Newton increment of every iteration moved by <= cond(H) eps (in the Newton loop below).
 * Modes 1-3 add exactly the same terms: what differs is rounding in the last
 * bits of the 28 sums.  tests/test_gpu_fullsize.py uses it to MEASURE which registrations are chaotic (their control flow
 * changes with the summation order alone) instead of asserting it. */
static int g_sum_mode = 0;
void oracle_set_sum_mode(int mode) { g_sum_mode = mode; }

static double derivatives_cells_impl(const oracle_map *target, const ocell *src, size_t msrc, int n_neighbours,
                                     int with_hessian, double lfd1, double lfd2, double g[6], double H[36])
{
    double score = 0;
    memset(g, 0, 6 * sizeof(double));
    memset(H, 0, 36 * sizeof(double));
    if (g_sum_mode == 0) {
        for (size_t i = 0; i < msrc; i++)
            derivatives_one_cell(target, &src[i], n_neighbours, with_hessian, lfd1, lfd2, &score, g, H);
        return score;
    }
    const int n_sh = (g_sum_mode == 2 || g_sum_mode == 3) ? 8 : 1, rev = (g_sum_mode == 1 || g_sum_mode == 3);
    for (int sh = 0; sh < n_sh; sh++) {
        double ps = 0, pg[6], pH[36];
        memset(pg, 0, sizeof pg);
        memset(pH, 0, sizeof pH);
        for (size_t k = 0; k < msrc; k++) {
            const size_t i = rev ? msrc - 1 - k : k;
            if ((int)(i % (size_t)n_sh) != sh) continue;
            derivatives_one_cell(target, &src[i], n_neighbours, with_hessian, lfd1, lfd2, &ps, pg, pH);
        }
        score += ps;
        for (int a = 0; a < 6; a++) g[a] += pg[a];
        for (int a = 0; a < 36; a++) H[a] += pH[a];
    }
    if (g_sum_mode >= 4 && g_sum_mode < 16) {
        /* modes 4..15: reference order + every sum moved by at most 8 ulp (a deterministic hash of mode and position, the
         * Hessian kept symmetric): the size of what ANY other arithmetic of the same formulas (another inverse, another exp,
         * fused multiply-adds) does to the sums */
        unsigned long long z = 0x9E3779B97F4A7C15ull * (unsigned long long)g_sum_mode;
        double *v[1 + 6 + 21];
        int n = 0;
        v[n++] = &score;
        for (int a = 0; a < 6; a++) v[n++] = &g[a];
        for (int a = 0; a < 6; a++)
            for (int b = a; b < 6; b++) v[n++] = &H[a * 6 + b];
        for (int k = 0; k < n; k++) {
            z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
            const int e = (int)(z % 17ull) - 8;                                 /* -8 .. 8 ulp */
            *v[k] *= 1.0 + (double)e * 2.220446049250313e-16;
        }
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < a; b++) H[a * 6 + b] = H[b * 6 + a];
    }
    return score;
}
/* (ORACLE_TRACE in the environment: one line per derivative evaluation on stderr -- debugging aid for parity work) */
static double derivatives_cells(const oracle_map *target, const ocell *src, size_t msrc, int n_neighbours,
                                int with_hessian, double lfd1, double lfd2, double g[6], double H[36])
{
    const double s = derivatives_cells_impl(target, src, msrc, n_neighbours, with_hessian, lfd1, lfd2, g, H);
    if (getenv("ORACLE_TRACE"))
        fprintf(stderr, "oracle eval with_h %d score %.17g g %.9e %.9e %.9e\n", with_hessian, s, g[0], g[1], g[5]);
    return s;
}

/* NDTMatcherFeatureD2D::derivativesNDT (perception_oru ndt_registration, restated from memory): the D2D pair term of
 * derivativesNDT above for the KNOWN correspondences corr[i] = (i, i) of two CellVector maps -- no neighbourhood search.
 * Call sites: [fusion.h]:858, 1087 and, through the virtual call inside lineSearchMT, :1016. */
static double derivatives_feat(const ocell *tgt, const ocell *src, size_t n, int with_hessian, double lfd1, double lfd2,
                               double g[6], double H[36])
{
    double score = 0;
    memset(g, 0, 6 * sizeof(double));
    memset(H, 0, 36 * sizeof(double));
    for (size_t i = 0; i < n; i++) {
        vec3 mm = {{src[i].mean[0], src[i].mean[1], src[i].mean[2]}};
        mat3 CM = src[i].cov;
        local_derivs L;
        compute_derivatives_local(mm, CM, with_hessian, &L);
        const ocell *tc = &tgt[i];
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

static int invert6(const double *A, double *inv);

/* NDTMatcherD2D::covariance(target, source, T, cov)  (ndt_feature_graph.cpp:296-298; fuser_hmt.cpp:403-405).
 * perception_oru, restated from memory (SURVEY App. A.7 -- the least certain part of the path):
 *   cov = H^-1 (sigma_S J^T J) H^-1,  sigma_S = 0.03^2,
 * H = the D2D Hessian of derivativesNDT at T, J = one row per source cell whose (transformed) mean falls into a
 * target cell with a Gaussian:
 *   x = m_src - m_tgt,  B = (C_tgt + C_src)^-1,  f = exp(lfd2 * (-x^T B x / 2)) / 2  (skipped when x^T B x / 2 > 120
 *   or f outside [0, 1]),  Q = -sigma_S B B,
 *   G_a = x^T Q j_a  [- x^T Q Z_a B x - x^T B Z_a Q x for the rotations]  + (-lfd2 / 2) x^T Q x,
 *   row = G * f * lfd1 * lfd2 / 2.
 * mode 0: j_a, Z_a are the derivatives of THIS source cell (computeDerivativesLocal); mode 1: the matcher's member
 * copies as its constructor leaves them (j = [I 0], Z = 0) -- what the OpenMP revisions, whose derivativesNDT works on
 * thread-local copies, effectively use.  The reference's extra rows for target cells whose transformed mean equals
 * the matched mean only exist when T is the identity and are not reproduced. */
int oracle_covariance(const oracle_map *target, const oracle_map *source, const double T[16], int n_neighbours,
                      double lfd1, double lfd2, int mode, double cov36[36])
{
    const double sigmaS = 0.03 * 0.03;
    size_t n;
    ocell *src = pseudo_transform(source, T, &n);
    double g6[6], H[36], JK[36];
    derivatives_cells(target, src, n, n_neighbours, 1, lfd1, lfd2, g6, H);
    memset(JK, 0, sizeof JK);
    for (size_t i = 0; i < n; i++) {
        int idx[3];
        index_for_point(target, src[i].mean, idx);
        if (!idx_inside(target, idx)) continue;
        int32_t c = target->cell_of_slot[slot_of(target, idx)];
        if (c < 0 || !target->cells[c].has_gaussian) continue;
        const ocell *tc = &target->cells[c];
        vec3 x = {{src[i].mean[0] - tc->mean[0], src[i].mean[1] - tc->mean[1], src[i].mean[2] - tc->mean[2]}};
        mat3 B;
        double det;
        if (!m3_inverse_check(m3_add(tc->cov, src[i].cov), &B, &det)) continue;
        double factor = -v3_dot(x, m3_v(B, x)) / 2;
        if (factor < -120) continue;
        factor = exp(lfd2 * factor) / 2;
        if (factor > 1 || factor < 0 || factor * 0 != 0) continue;
        mat3 Q = m3_mul(B, B);
        for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++) Q.m[r][q] *= -sigmaS;
        local_derivs L;
        if (mode == 0) {
            vec3 mm = {{src[i].mean[0], src[i].mean[1], src[i].mean[2]}};
            compute_derivatives_local(mm, src[i].cov, 0, &L);
        } else {
            memset(&L, 0, sizeof L);
            for (int a = 0; a < 3; a++) L.J[a].v[a] = 1.0;
        }
        vec3 xQ = m3_v(m3_T(Q), x), xB = m3_v(m3_T(B), x), Bx = m3_v(B, x), Qx = m3_v(Q, x);
        double f1 = v3_dot(xQ, x), G[6];
        for (int a = 0; a < 6; a++) {
            double ga = 0;
            if (a >= 3) {
                ga = -v3_dot(xQ, m3_v(L.Z[a], Bx));
                ga = ga - v3_dot(xB, m3_v(L.Z[a], Qx));
            }
            G[a] = (ga + v3_dot(xQ, L.J[a]) + (-lfd2 / 2) * f1) * factor * lfd1 * lfd2 / 2;
        }
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < 6; b++) JK[a * 6 + b] += G[a] * G[b];
    }
    free(src);
    for (int a = 0; a < 36; a++) JK[a] *= sigmaS;
    double Hinv[36], tmp[36];
    if (!invert6(H, Hinv)) return -2;
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) {
            double s2 = 0;
            for (int k = 0; k < 6; k++) s2 += Hinv[a * 6 + k] * JK[k * 6 + b];
            tmp[a * 6 + b] = s2;
        }
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) {
            double s2 = 0;
            for (int k = 0; k < 6; k++) s2 += tmp[a * 6 + k] * Hinv[k * 6 + b];
            cov36[a * 6 + b] = s2;
        }
    return 0;
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

typedef struct {
    const ocell *tgt;   /* targetNDT_feat */
    const ocell *cells; /* nextNDT_feat */
    ocell *scratch;
    size_t n;
    const double *incr;
    const oracle_match_params *prm;
} lsf_ctx;

static double lsf_phi(void *vctx, double stp, double *dg)
{
    lsf_ctx *c = (lsf_ctx *)vctx;
    double pincr[6], ps[16], g[6], H[36];
    for (int a = 0; a < 6; a++) pincr[a] = stp * c->incr[a];
    oracle_pose_to_T(pincr, ps);
    transform_cells(c->cells, c->scratch, c->n, ps);
    double f = derivatives_feat(c->tgt, c->scratch, c->n, 0, c->prm->lfd1, c->prm->lfd2, g, H);
    double d = 0;
    for (int a = 0; a < 6; a++) d += c->incr[a] * g[a];
    *dg = d;
    return f;
}

/* NDTMatcherFeatureD2D::lineSearchMT = NDTMatcherD2D::lineSearchMT with the feature derivatives ([fusion.h]:1016) */
static double line_search_mt_feat(double incr[6], const ocell *tgt, const ocell *cells, ocell *scratch, size_t n,
                                  const oracle_match_params *prm)
{
    double g[6], H[36];
    double score_init = derivatives_feat(tgt, cells, n, 0, prm->lfd1, prm->lfd2, g, H);
    double dginit = 0;
    for (int a = 0; a < 6; a++) dginit += incr[a] * g[a];
    if (dginit >= 0.0) {
        for (int a = 0; a < 6; a++) incr[a] = -incr[a];
        dginit = -dginit;
        if (dginit >= 0.0) return 0.1; /* recoverystep */
    }
    lsf_ctx ctx = {tgt, cells, scratch, n, incr, prm};
    return oracle_mt_linesearch(lsf_phi, &ctx, score_init, dginit, NULL, NULL);
}

/* lineSearchMTFusion ([fusion.h]:390-793): ONE More-Thuente search on the sum of the NDT and the feature function.  As
 * written upstream the feature maps are evaluated on the UN-stepped cells in every trial ([fusion.h]:619 passes
 * sourceNDT_feat, not sourceNDTHere_feat): their score and gradient at the current pose are constants of the search.
 * Restated as written, not "fixed". */
typedef struct {
    ls_ctx ndt;
    double f_feat, g_feat[6];
} lsj_ctx;

static double lsj_phi(void *vctx, double stp, double *dg)
{
    lsj_ctx *c = (lsj_ctx *)vctx;
    double pincr[6], ps[16], g[6], H[36];
    for (int a = 0; a < 6; a++) pincr[a] = stp * c->ndt.incr[a];
    oracle_pose_to_T(pincr, ps);
    transform_cells(c->ndt.cells, c->ndt.scratch, c->ndt.n, ps);
    double f = derivatives_cells(c->ndt.target, c->ndt.scratch, c->ndt.n, c->ndt.prm->n_neighbours, 0, c->ndt.prm->lfd1,
                                 c->ndt.prm->lfd2, g, H);
    (*c->ndt.fevals)++;
    double d = 0;
    for (int a = 0; a < 6; a++) d += c->ndt.incr[a] * (g[a] + c->g_feat[a]);
    *dg = d;
    return f + c->f_feat;
}

static double line_search_mt_fusion(double incr[6], const oracle_map *target, const ocell *cells, ocell *scratch, size_t n,
                                    const ocell *ftgt, const ocell *fcells, size_t nf, const oracle_match_params *prm, int *fevals)
{
    double g[6], H[36], gf[6], Hf[36];
    double score_init = derivatives_cells(target, cells, n, prm->n_neighbours, 0, prm->lfd1, prm->lfd2, g, H);
    (*fevals)++;
    const double ff = derivatives_feat(ftgt, fcells, nf, 0, prm->lfd1, prm->lfd2, gf, Hf);
    score_init += ff;
    double dginit = 0;
    for (int a = 0; a < 6; a++) dginit += incr[a] * (g[a] + gf[a]);
    if (dginit >= 0.0) {
        for (int a = 0; a < 6; a++) incr[a] = -incr[a];
        dginit = -dginit;
        if (dginit >= 0.0) return 0.1; /* recoverystep */
    }
    lsj_ctx ctx;
    ls_ctx nd = {target, cells, scratch, n, incr, prm, fevals};
    ctx.ndt = nd;
    ctx.f_feat = ff;
    memcpy(ctx.g_feat, gf, sizeof gf);
    return oracle_mt_linesearch(lsj_phi, &ctx, score_init, dginit, NULL, NULL);
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

typedef struct {   /* the feature / odometry-cell maps of matchFusion: n cells each, corr_feat[i] = (i, i) */
    size_t n;
    const double *src_mean, *src_cov6, *tgt_mean, *tgt_cov6;
} feat_maps;
static int match_common(const oracle_map *target, const oracle_map *source, double T[16],
                        const oracle_match_params *prm, const double *Q /* Tcov^-1 or NULL */, int flags,
                        const feat_maps *feat, oracle_match_result *res);

/* test aid: how often the in-place negation of lineSearchMTFusionTcov ([fusion.h]:89-95) has fired since the last reset */
static long g_tcov_flips = 0;
long oracle_debug_tcov_flips(int reset)
{
    long v = g_tcov_flips;
    if (reset) g_tcov_flips = 0;
    return v;
}

int oracle_match_d2d(const oracle_map *target, const oracle_map *source, double T[16],
                     const oracle_match_params *prm, oracle_match_result *res)
{
    return match_common(target, source, T, prm, NULL, 0, NULL, res);
}

/* ndt_feature::matchFusion ([fusion.h]:797-1155) with useNDT = true, useFeat = false (empty feature
 * maps: their derivativesNDT terms are zero), useTikhonovRegularization = false.  With
 * useSoftConstraints the score / gradient / Hessian get the Mahalanobis terms of X = pose_local_v
 * ([fusion.h]:875-890, 1098-1110); the step length comes from NDTMatcherD2D::lineSearchMT because the
 * result of lineSearchMTFusionTcov is overwritten ([fusion.h]:1008-1023: step_size_feat == 0 ->
 * step_size = max(step_size_ndt, 0)); what survives of that discarded search is the in-place negation of the
 * increment when increment . (g_ndt + g_mahalanobis) >= 0 ([fusion.h]:89-95), restated in match_common. */
int oracle_match_fusion(const oracle_map *target, const oracle_map *source, double T[16],
                        const oracle_match_params *prm, const double Tcov[36], int use_soft_constraints,
                        oracle_match_result *res)
{
    /* use_soft_constraints: bit 0 = useSoftConstraints, bit 1 = useTikhonovRegularization ([fusion.h]:894-911) */
    double Q[36];
    if (!(use_soft_constraints & 3)) return match_common(target, source, T, prm, NULL, 0, NULL, res);
    if (!invert6(Tcov, Q)) return -2;
    return match_common(target, source, T, prm, Q, use_soft_constraints & 3, NULL, res);
}

/* ndt_feature::matchFusion with useFeat ([fusion.h]:797-1155): n_feat correspondences between sourceNDT_feat and
 * targetNDT_feat (cov6: xx xy xz yy yz zz).  flags: bit 0 useSoftConstraints, bit 1 useTikhonovRegularization.  The
 * separate line searches of [fusion.h]:1007-1023, or (bit 2 step_control_fusion set, bit 0 clear) the joint
 * lineSearchMTFusion of :1004-1006.  res->fevals counts NDT-map evaluations. */
int oracle_match_fusion_feat(const oracle_map *target, const oracle_map *source, double T[16],
                             const oracle_match_params *prm, const double Tcov[36], int flags, size_t n_feat,
                             const double *src_mean, const double *src_cov6, const double *tgt_mean, const double *tgt_cov6,
                             oracle_match_result *res)
{
    double Q[36];
    feat_maps f = {n_feat, src_mean, src_cov6, tgt_mean, tgt_cov6};
    if (flags & 3) {
        if (!invert6(Tcov, Q)) return -2;
        return match_common(target, source, T, prm, Q, flags & 7, &f, res);
    }
    return match_common(target, source, T, prm, NULL, flags & 4, &f, res);
}

/* x0 = convertAffineToVector(forceEigenAffine3dTo2d(T * Tinit^-1))  ([fusion.h]:903-907, utils.h:30-68, 161-169) */
static void tikhonov_x0(const double T[16], const double Tinit[16], double x0[6])
{
    double Ti[16], D[16];
    memset(Ti, 0, sizeof Ti);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) Ti[c * 4 + r] = Tinit[r * 4 + c];
        Ti[12 + r] = -(Tinit[r * 4 + 0] * Tinit[12] + Tinit[r * 4 + 1] * Tinit[13] + Tinit[r * 4 + 2] * Tinit[14]);
    }
    Ti[15] = 1.0;
    T_mul(T, Ti, D);
    double c = D[0];
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);      /* getRobustYawFromAffine3d: acos(dot), sign from the cross product */
    double ang = acos(c);
    x0[0] = D[12]; x0[1] = D[13]; x0[2] = 0;
    x0[3] = 0; x0[4] = 0; x0[5] = (D[1] > 0) ? ang : -ang;
}

static void feat_cell(ocell *c, const double *mean, const double *cov6)
{
    memset(c, 0, sizeof *c);
    for (int a = 0; a < 3; a++) c->mean[a] = mean[a];
    c->cov.m[0][0] = cov6[0]; c->cov.m[0][1] = c->cov.m[1][0] = cov6[1]; c->cov.m[0][2] = c->cov.m[2][0] = cov6[2];
    c->cov.m[1][1] = cov6[3]; c->cov.m[1][2] = c->cov.m[2][1] = cov6[4]; c->cov.m[2][2] = cov6[5];
    c->has_gaussian = 1;
}

static int match_common(const oracle_map *target, const oracle_map *source, double T[16],
                        const oracle_match_params *prm, const double *Q, int flags, const feat_maps *feat,
                        oracle_match_result *res)
{
    const int soft = Q && (flags & 1), tikhonov = Q && (flags & 2);
    const size_t nf = feat ? feat->n : 0;
    /* a registration without correspondences runs with useFeat = false: the reference's call site never passes useFeat = true
     * with empty feature maps (ndt_feature_fuser_hmt.cpp:296-320, 341-347); what the feature line search would do on empty
     * maps (dginit = 0: negated increment, recovery step) is not restated -- include/ndtgpu.h says the same of the HIP path */
    const int use_feat = nf > 0;
    const int joint_ls = (flags & 4) && !soft;      /* useNDT && useFeat && step_control_fusion && !useSoftConstraints */
    double Tinit[16], x0[6] = {0, 0, 0, 0, 0, 0};
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
    memcpy(Tinit, T, sizeof Tinit);

    size_t n;
    double pose_local[6] = {0, 0, 0, 0, 0, 0};
    ocell *next = pseudo_transform(source, T, &n); /* [fusion.h]:840 */
    ocell *scratch = (ocell *)calloc(n ? n : 1, sizeof(ocell));
    double g6[6], H36[36];
    /* nextNDT_feat = sourceNDT_feat.pseudoTransformNDT(T) ([fusion.h]:841); targetNDT_feat as given */
    ocell *ftgt = (ocell *)calloc(nf ? nf : 1, sizeof(ocell)), *fnext = (ocell *)calloc(nf ? nf : 1, sizeof(ocell)),
          *fscratch = (ocell *)calloc(nf ? nf : 1, sizeof(ocell));
    for (size_t i = 0; i < nf; i++) {
        feat_cell(&ftgt[i], feat->tgt_mean + 3 * i, feat->tgt_cov6 + 6 * i);
        feat_cell(&fnext[i], feat->src_mean + 3 * i, feat->src_cov6 + 6 * i);
    }
    if (nf) transform_cells(fnext, fnext, nf, T);

    while (!convergence) {
        score_here = derivatives_cells(target, next, n, prm->n_neighbours, 1, prm->lfd1, prm->lfd2, g6, H36);
        fevals++;
        if (use_feat) { /* [fusion.h]:858-871 */
            double gf[6], Hf[36];
            score_here += derivatives_feat(ftgt, fnext, nf, 1, prm->lfd1, prm->lfd2, gf, Hf);
            for (int i = 0; i < 6; i++) g6[i] += gf[i];
            for (int i = 0; i < 36; i++) H36[i] += Hf[i];
        }
        if (soft) { /* [fusion.h]:875-890 */
            double gq[6], Hq[36];
            score_here += oracle_mahalanobis(pose_local, Q, gq, Hq);
            for (int i = 0; i < 6; i++) g6[i] += gq[i];
            for (int i = 0; i < 36; i++) H36[i] += Hq[i];
        }
        /* what lineSearchMTFusionTcov evaluates at the current pose ([fusion.h]:77-89): NDT + Mahalanobis gradient,
         * BEFORE the Tikhonov transformation */
        double g_tcov[6];
        memcpy(g_tcov, g6, sizeof g_tcov);
        if (tikhonov) { /* [fusion.h]:894-911, P = I */
            double g2[6], H2[36];
            tikhonov_x0(T, Tinit, x0);
            for (int a = 0; a < 6; a++) {
                double s1 = 0;
                for (int k = 0; k < 6; k++) s1 += H36[k * 6 + a] * g6[k] + Q[a * 6 + k] * x0[k];
                g2[a] = s1;
                for (int b = 0; b < 6; b++) {
                    double s2 = Q[a * 6 + b];
                    for (int k = 0; k < 6; k++) s2 += H36[k * 6 + a] * H36[k * 6 + b];
                    H2[a * 6 + b] = s2;
                }
            }
            memcpy(g6, g2, sizeof g2);
            memcpy(H36, H2, sizeof H2);
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) score_here += x0[i] * Q[i * 6 + j] * x0[j];
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
        if (g_sum_mode >= 16) {
            /* modes 16..: the Newton increment moved by up to cond(H) eps per component -- the forward error of ANY
             * backward-stable solve of H dx = g, i.e. what another implementation of the same solve (fused multiply-adds,
             * a reciprocal square root in the Cholesky, another pivot order) does to it.  cond(H) from the eigenvalues the
             * regulariser just computed (|lambda|max / |lambda|min of the matrix that is solved). */
            double lo = fabs(ev[0]), hi = fabs(ev[0]);
            for (int i = 1; i < nd; i++) { const double a = fabs(ev[i]); lo = a < lo ? a : lo; hi = a > hi ? a : hi; }
            const double kappa = lo > 0 ? hi / lo : 1e16;
            /* bounded: a (near-)singular H must not turn this into an O(1) change of the step, behind which a real
             * discrepancy of the implementation under test could hide -- at most 1e-9 relative */
            double amp = kappa * 2.220446049250313e-16;
            if (!(amp < 1e-9)) amp = 1e-9;
            unsigned long long z = 0xD1B54A32D192ED03ull * (unsigned long long)(g_sum_mode + 31 * itr_ctr);
            for (int i = 0; i < nd; i++) {
                z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
                dx[i] *= 1.0 + ((double)((int)(z % 17ull) - 8) / 8.0) * amp;
            }
        }
        double dginit = 0;
        for (int i = 0; i < nd; i++) {
            incr[dofs[i]] = -dx[i];
            dginit += -dx[i] * g[i];
        }
        if (getenv("ORACLE_TRACE"))
            fprintf(stderr, "oracle it %d score %.17g gnorm %.6e minC %.6e maxC %.6e dginit %.6e g %.6e %.6e %.6e\n", itr_ctr,
                    score_here, gnorm, minC, maxC, dginit, g[0], g[1], nd > 2 ? g[2] : 0.0);
        if (dginit > 0) {
            if (score_here > score_best) memcpy(T, Tbest, sizeof Tbest);
            exit_code = 2;
            goto done_early;
        }
        /* [fusion.h]:1000-1032 */
        double step_size = 1.0;
        if (prm->step_control) {
            /* [fusion.h]:1008-1010: with useSoftConstraints lineSearchMTFusionTcov runs first.  Its step is overwritten
             * ([fusion.h]:1018-1023: step_size_feat == 0 -> step_size = max(step_size_ndt, 0)), but it takes the increment
             * BY REFERENCE and negates it in place when increment . (g_ndt + g_mahalanobis) >= 0 ([fusion.h]:89-95);
             * lineSearchMT then starts from the negated vector.  Without Tikhonov that gradient is scg, on which
             * dginit <= 0 was just established (only dginit == 0 flips); with Tikhonov scg = H^T g + Q x0 is another
             * vector and the flip is reachable.  The discarded search itself is not evaluated. */
            if (soft) {
                double dtcov = 0;
                for (int a = 0; a < 6; a++) dtcov += incr[a] * g_tcov[a];
                if (dtcov >= 0.0) {
                    for (int a = 0; a < 6; a++) incr[a] = -incr[a];
                    g_tcov_flips++;
                }
            }
            /* the line search sees only the active dofs' gradient through incr (inactive entries are 0) */
            if (use_feat && joint_ls) /* [fusion.h]:1004-1006 */
                step_size = line_search_mt_fusion(incr, target, next, scratch, n, ftgt, fnext, nf, prm, &fevals);
            else
                step_size = line_search_mt(incr, target, next, scratch, n, prm, &fevals);
            if (use_feat && !joint_ls) { /* [fusion.h]:1015-1023 */
                double step_size_feat = line_search_mt_feat(incr, ftgt, fnext, fscratch, nf, prm);
                if (step_size != 0. && step_size_feat != 0.) step_size = step_size < step_size_feat ? step_size : step_size_feat;
                else step_size = step_size > step_size_feat ? step_size : step_size_feat;
            }
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
        if (nf) transform_cells(fnext, fnext, nf, TR);
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
    if (use_feat) { /* [fusion.h]:1087-1096 */
        double gf[6], Hf[36];
        score_here += derivatives_feat(ftgt, fnext, nf, 0, prm->lfd1, prm->lfd2, gf, Hf);
    }
    if (soft) { /* [fusion.h]:1098-1110 */
        double gq[6], Hq[36];
        score_here += oracle_mahalanobis(pose_local, Q, gq, Hq);
    }
    if (tikhonov) /* [fusion.h]:1113-1115: the x0 of the last loop pass */
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) score_here += x0[i] * Q[i * 6 + j] * x0[j];
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
    free(ftgt);
    free(fnext);
    free(fscratch);
    return 0;
}
