/*
 * ndt_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Single-threaded fp64 plain-C restatement of the NDT hot path that
 * MalcolmMielle/ndt_feature_graph drives through perception_oru
 * (ndt_map / ndt_registration, un-vendored, NO version pinned anywhere in the
 * reference: ndt_feature/package.xml:34-45).
 *
 * PARITY UNPINNED: the reference holds no golden vector, known-answer test or
 * fixture for this path and cannot be compiled here
 * (needs Eigen, PCL, Boost, ROS, perception_oru).  The restatement is anchored
 * on the in-repo verbatim-derived copy of the Newton loop and More-Thuente
 * driver (ndt_feature/include/ndt_feature/ndt_matcher_d2d_fusion.h:390-793,
 * 797-1155), the reference call sites, the published MINPACK-2 dcstep
 * algorithm, and first-principles derivation of the D2D derivatives that is
 * checked by finite differences in tests/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use this library, and only as the checker / the timed CPU baseline.
 */
#ifndef NDT_ORACLE_H
#define NDT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_map oracle_map;

/* lslgeneric::LazyGrid(res) + NDTMap::initialize / guessSize
 * (ndt_feature_fuser_hmt.cpp:87-89, 195-196, 222). */
oracle_map *oracle_map_create(double res, const double centre[3], const double size_m[3]);
void oracle_map_destroy(oracle_map *m);

/* LazyGrid::getIndexForPoint: idx = floor((p-c)/res + 0.5) + size/2.0 -> int.
 * Returns 1 when inside the grid. */
int oracle_map_index_for_point(const oracle_map *m, const double p[3], int idx[3]);

/* NDTMap::loadPointCloud(pc, range_limit) (ndt_feature_fuser_hmt.cpp:225) and
 * the point-binning half of loadPointCloudCentroid (:201,216) when
 * range_origin != NULL (range measured from that origin).
 * Replaces the map content.  stride in floats (3 = packed xyz, 4 = PointXYZ). */
int oracle_map_load_points(oracle_map *m, const float *xyz, size_t n, size_t stride_floats,
                           double range_limit, const double *range_origin);

/* NDTMap::computeNDTCells(CELL_UPDATE_MODE_SAMPLE_VARIANCE) ->
 * NDTCell::computeGaussian (first Gaussian) + rescaleCovariance
 * (ndt_feature_fuser_hmt.cpp:227; ndt_odom_debug.cpp:179). */
int oracle_map_compute_cells(oracle_map *m, int n_min, double eval_factor);

/* NDTMap::computeNDTCells(SAMPLE_VARIANCE, maxnumpoints, occupancy_limit, origin, sensor_noise)
 * (ndt_feature_fuser_hmt.cpp:94, 486): occupancy log-odds, first Gaussian or recursive (N, mean, cov) merge with
 * the `maxnumpoints` saturation, occ <= 0 => no Gaussian, rescaleCovariance (perception_oru). */
int oracle_map_compute_cells_full(oracle_map *m, int n_min, double eval_factor, double maxnumpoints,
                                  double occupancy_limit);

/* NDTMap::addPointCloud(origin, cloud, classifierTh, maxz, sensor_noise, occupancy_limit)
 * (ndt_feature_fuser_hmt.cpp:92, 485) on an initialize()d map: per beam, the cells between sensor and hit get
 * emptiness evidence, the hit joins its cell.  order_free: 0 = the reference's beam-after-beam semantics,
 * 1 = the order-independent semantics of the HIP path (see ndt_oracle.c). */
int oracle_map_add_point_cloud(oracle_map *m, const double origin[3], const float *xyz, size_t n,
                               size_t stride_floats, double maxz, double sensor_noise, double occupancy_limit,
                               int order_free);
/* one beam x one Gaussian cell: the log-odds update (float) or 0 = cell untouched */
int oracle_beam_evidence(const double mean[3], const double cov9[9], const double origin[3], const float pe[3],
                         double sensor_noise, float *logodd);
/* cells per axis */
void oracle_map_size(const oracle_map *m, int size[3]);
/* NDTCell::occ of every slot (x-major, y, z) */
void oracle_map_occupancy(const oracle_map *m, float *occ_out);
/* NDTCell::getOccupancyRescaled */
float oracle_occupancy_rescaled(float occ);
/* ndt_feature::overlapNDTOccupancyScore(ref, mov, T) (ndt_feature_node.h:213-252); T column-major 4x4.
 * *nb_sum = number of cell pairs compared. */
double oracle_overlap_score(const oracle_map *ref, const oracle_map *mov, const double T[16], long long *nb_sum);

/* number of cells with hasGaussian_, in slot order (x-major, then y, then z) */
int oracle_map_num_cells(const oracle_map *m);
/* export gaussian cells in slot order: mean3[3*i], cov9[9*i] (row-major), idx3, npts */
int oracle_map_export_cells(const oracle_map *m, double *mean3, double *cov9, int32_t *idx3,
                            int32_t *npts);
/* install cells directly (CellVector::addNDTCell-like; used by KATs) */
int oracle_map_set_cells(oracle_map *m, const double *mean3, const double *cov9, size_t ncells);

typedef struct {
    int n_neighbours;   /* matcher_d2d.n_neighbours (ndt_feature_graph.cpp:262) */
    int itr_max;        /* ITR_MAX (ndt_feature_fuser_hmt.h:82) */
    double delta_score; /* DELTA_SCORE */
    int step_control;   /* More-Thuente on/off */
    double lfd1, lfd2;  /* 1, 0.05 */
    int dof_mask;       /* bit a set = pose dof a active; 0x3f = 6-DoF, 0x23 = {x,y,yaw} */
    int use_initial_guess;
} oracle_match_params;

typedef struct {
    int converged;   /* ret of match(): 0 when the iteration cap was hit */
    int iterations;  /* itr_ctr at exit */
    int fevals;      /* number of derivativesNDT evaluations */
    double score;    /* final score (after best-score rollback: score at returned T is min(best,final)) */
    int exit_code;   /* 0 step<delta, 1 gradient vanished, 2 wrong direction, 3 iteration cap */
} oracle_match_result;

/* NDTMatcherD2D::derivativesNDT (called at ndt_matcher_d2d_fusion.h:856,444,617,1085).
 * src cells are already in the target frame.  g[6], H[36] row-major. */
double oracle_derivatives(const oracle_map *target, const double *src_mean3, const double *src_cov9,
                          size_t m, int n_neighbours, int compute_hessian, double lfd1, double lfd2,
                          double g[6], double H[36]);

/* D2D score only at pose increment p applied on the left of T (for FD tests):
 * cells transformed by TR(p)*T then summed. */
double oracle_score_at(const oracle_map *target, const oracle_map *source, const double T[16],
                       const double p[6], int n_neighbours, double lfd1, double lfd2);

/* NDTMatcherD2D::match(target, source, T, useInitialGuess) (ndt_feature_graph.cpp:273),
 * loop restated from ndt_matcher_d2d_fusion.h:847-1121 with the NDT term only.
 * T: 4x4 column-major (Eigen::Affine3d storage), in/out. */
int oracle_match_d2d(const oracle_map *target, const oracle_map *source, double T[16],
                     const oracle_match_params *prm, oracle_match_result *res);

/* ndt_feature::matchFusion (ndt_matcher_d2d_fusion.h:797-1155) with empty feature maps and without the
 * Tikhonov variant: NDT term + (optionally) the odometry soft constraint x^T Tcov^-1 x.
 * Tcov: 6x6 row-major.  Returns -2 when Tcov is singular. */
/* ndt_feature::matchFusion with useFeat: n_feat correspondences (i <-> i) between the cells of sourceNDT_feat and
 * targetNDT_feat (cov6: xx xy xz yy yz zz); flags bit 0 useSoftConstraints, bit 1 useTikhonovRegularization */
int oracle_match_fusion_feat(const oracle_map *target, const oracle_map *source, double T[16],
                             const oracle_match_params *prm, const double Tcov[36], int flags, size_t n_feat,
                             const double *src_mean, const double *src_cov6, const double *tgt_mean, const double *tgt_cov6,
                             oracle_match_result *res);
/* test aid: number of in-place negations of the increment by lineSearchMTFusionTcov ([fusion.h]:89-95) so far */
long oracle_debug_tcov_flips(int reset);
int oracle_match_fusion(const oracle_map *target, const oracle_map *source, double T[16],
                        const oracle_match_params *prm, const double Tcov[36], int use_soft_constraints,
                        oracle_match_result *res);

/* NDTMatcherD2D::covariance(target, source, T, cov) (ndt_feature_graph.cpp:296-298): cov = H^-1 (0.03^2 J^T J) H^-1;
 * mode 0 = per-cell Jacobians, 1 = the matcher's constructor values (see ndt_oracle.c).  cov36 row-major.
 * Returns -2 when the Hessian is singular. */
int oracle_covariance(const oracle_map *target, const oracle_map *source, const double T[16], int n_neighbours,
                      double lfd1, double lfd2, int mode, double cov36[36]);

/* MoreThuente::cstep (MINPACK-2 dcstep; called at ndt_matcher_d2d_fusion.h:756,775). */
int oracle_mt_cstep(double *stx, double *fx, double *dx, double *sty, double *fy, double *dy,
                    double *stp, double fp, double dp, int *brackt, double stmin, double stmax);

/* The More-Thuente driver of ndt_matcher_d2d_fusion.h:390-793 on a scalar
 * function phi(stp) -> (f, dg) supplied by the caller; finit/dginit are
 * phi(0).  Returns the step (recovery step 0.1 on failure); *nfev_out =
 * number of phi evaluations. */
typedef double (*oracle_phi_fn)(void *ctx, double stp, double *dg);
double oracle_mt_linesearch(oracle_phi_fn phi, void *ctx, double finit, double dginit,
                            int *nfev_out, int *info_out);

/* small algebra exported for KATs */
void oracle_pose_to_T(const double p[6], double T[16]);          /* Trans*Rx*Ry*Rz (fusion.h:1036-1039) */
int oracle_eig_sym(int n, const double *A, double *evals, double *evecs); /* n<=6, row-major */
int oracle_ldlt_solve(int n, const double *A, const double *b, double *x);
/* computeScore/Gradient/HessianMahalanobis (ndt_matcher_d2d_fusion.h:11-32) */
double oracle_mahalanobis(const double x[6], const double Q[36], double g[6], double H[36]);

/* test knob: order in which derivativesNDT adds the source cells' terms (0 reference order, 1 reversed, 2 eight shares
 * i mod 8 added in share order, 3 eight shares reversed inside); same terms, other rounding */
void oracle_set_sum_mode(int mode);

#ifdef __cplusplus
}
#endif
#endif
