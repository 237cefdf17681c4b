/*
 * ndtgpu.h -- C-ABI of the MI355X-native NDT scan-matching front-end.
 *
 * Drop-in boundary for the lslgeneric:: classes that MalcolmMielle/ndt_feature_graph calls
 * on its hot path (there is no FFI layer in the reference: the path sits behind the C++
 * class API of perception_oru's ndt_map / ndt_registration).  Each
 * entry point cites the reference interface it replaces (paths relative to the reference
 * root).  Plain pointers and sizes only; handles are opaque; outputs are caller-owned.
 *
 * Conventions
 *   - 4x4 poses are 16 doubles, COLUMN-major, exactly Eigen::Affine3d::data().
 *   - points are float xyz records, `stride_bytes` apart (12 = packed, 16 = pcl::PointXYZ).
 *   - a "mapset" is B NDT maps that share one grid geometry (cell size, extent in cells) and
 *     live in one device arena; a single lslgeneric::NDTMap is a mapset with B = 1.
 *   - every call returns ndtgpu_status; convergence is reported in the result struct, never
 *     as an error (reference: match() returns bool, fusion.h:1075-1079).
 *   - handles are not thread-safe; distinct handles may be used from distinct threads.
 *   - the library never falls back to the CPU: without a HIP device every compute entry
 *     point returns NDTGPU_ERR_NO_DEVICE.
 */
#ifndef NDTGPU_H
#define NDTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int ndtgpu_status;
enum {
    NDTGPU_OK = 0,
    NDTGPU_ERR_INVALID = -1,   /* bad argument */
    NDTGPU_ERR_HIP = -2,       /* HIP runtime error (see ndtgpu_last_error) */
    NDTGPU_ERR_NO_DEVICE = -3, /* no gfx950 device visible */
    NDTGPU_ERR_CAPACITY = -4,  /* a map needed more cells than max_cells */
    NDTGPU_ERR_ALLOC = -5
};

typedef struct ndtgpu_mapset ndtgpu_mapset;
typedef void *ndtgpu_stream; /* hipStream_t; NULL = default stream */

/* lslgeneric::LazyGrid(res) + NDTMap::initialize(cx,cy,cz,sx,sy,sz) / guessSize(...)
 * (ndt_feature/src/ndt_feature_src/ndt_feature_fuser_hmt.cpp:87-89, 195-196, 222). */
typedef struct {
    double res;        /* cubic cell size [m] (params_.resolution) */
    double centre[3];  /* grid centre [m]; per-map override: ndtgpu_mapset_set_centre */
    double size[3];    /* extent [m]; cells per axis = |ceil(size/res)| (LazyGrid::initialize) */
    uint32_t max_cells; /* capacity of occupied cells per map; 0 = min(slots, 16384) */
} ndtgpu_grid_params;

/* NDTCell::computeGaussian / rescaleCovariance knobs.
 * PROVENANCE: both constants live in perception_oru's ndt_map (un-vendored, no version pinned by the reference).
 *   n_min        the reference never passes it; the perception_oru revisions of the reference's era test
 *                `hasGaussian_ == false && points_.size() < 3` in the 5-argument computeGaussian that
 *                ndt_feature_fuser_hmt.cpp:94, 227, 486 reach (older / "simple" variants use 6).  Default 3;
 *                it decides which cells exist, so a deployment against a different perception_oru sets it here.
 *   eval_factor  EVAL_FACTOR of NDTCell; the reference itself passes 1000 when it builds cells by hand
 *                (ndt_feature/include/ndt_feature/utils.h:200). */
typedef struct {
    int32_t n_min;       /* points needed for a first Gaussian (default 3, see above) */
    double eval_factor;  /* EVAL_FACTOR, eigenvalue floor lambda_max/eval_factor (default 1000) */
} ndtgpu_cell_params;

/* NDTMatcherD2D members set by the callers (ndt_feature_graph.cpp:261-262;
 * ndt_feature_fuser_hmt.cpp:356-357; ndt_matcher_d2d_fusion.h:1170-1174).
 * PROVENANCE of the defaults (ndtgpu_default_match_params): the "fuser" preset -- what matchFusion is called with
 * in production: n_neighbours 2, ITR_MAX 30, DELTA_SCORE 1e-6 (launch/gustav_laser_tf.launch:56-59,
 * ndt_graph_offline.cpp:310-313), step_control on, lfd1 1, lfd2 0.05 (NDTMatcherD2D constructor, perception_oru).
 * The EDGE matcher of ndt_feature_graph.cpp:261 is default-constructed with only n_neighbours overridden; its
 * DELTA_SCORE is perception_oru's constructor value `10e-3 * current_resolution` with current_resolution = 0.1,
 * i.e. 1e-3 -- recalled, not readable in the reference: the host mirror
 * (host/lslgeneric_gpu.h NDTMatcherD2D::DELTA_SCORE) exposes it as a member like upstream does. */
typedef struct {
    int32_t n_neighbours;      /* matcher.n_neighbours */
    int32_t itr_max;           /* ITR_MAX */
    double delta_score;        /* DELTA_SCORE */
    int32_t step_control;      /* More-Thuente line search on/off */
    double lfd1, lfd2;         /* 1.0, 0.05 */
    int32_t dof_mask;          /* bit a = pose dof a active: 0x3f NDTMatcherD2D, 0x23 NDTMatcherD2D_2D */
    int32_t use_initial_guess; /* match(..., useInitialGuess) */
} ndtgpu_match_params;

typedef struct {
    int32_t converged;  /* return value of match(): 0 = iteration cap hit */
    int32_t iterations; /* itr_ctr at exit */
    int32_t fevals;     /* derivativesNDT evaluations */
    int32_t exit_code;  /* 0 step<delta, 1 gradient vanished, 2 wrong direction, 3 iteration cap;
                         * not run (device-pointer batches): -2 map index out of range, -3 a map overflowed max_cells,
                         * -4 the grid barrier of a small batch gave up (foreign work held CUs for seconds) */
    double score;       /* score at the returned pose */
    int32_t n_source;   /* Gaussian cells in the source map */
    int32_t n_target;
    int64_t cycles_eval;   /* shader clocks spent in derivative evaluations (profiling aid) */
    int64_t cycles_solver; /* shader clocks spent in the serial Newton / line-search code */
    int64_t pair_terms_g;  /* (source cell, target cell) terms summed in gradient-only evaluations */
    int64_t pair_terms_h;  /* ... and in evaluations with the Hessian (k-bar = terms / (fevals * n_source)) */
} ndtgpu_match_result;

/* ---- library ------------------------------------------------------------------------- */
const char *ndtgpu_version(void);
const char *ndtgpu_last_error(void);
/* number of usable devices (0 on a box without a GPU; never an error) */
int ndtgpu_device_count(void);
void ndtgpu_default_cell_params(ndtgpu_cell_params *p);
void ndtgpu_default_match_params(ndtgpu_match_params *p);

/* ---- maps ---------------------------------------------------------------------------- */
/* new NDTMap(new LazyGrid(res)) x n_maps + initialize()  (fuser_hmt.cpp:87-89, 195-196) */
ndtgpu_status ndtgpu_mapset_create(const ndtgpu_grid_params *grid, size_t n_maps, ndtgpu_mapset **out);
/* NDTMap destructor (ndt_feature_graph.h:78-88 deletes node maps) */
ndtgpu_status ndtgpu_mapset_destroy(ndtgpu_mapset *set);
/* LazyGrid::setCenter -- e.g. the snapped centroid of loadPointCloudCentroid
 * (fuser_hmt.cpp:201-217; ndt_odom_debug.cpp:191) or the node pose (fuser_hmt.cpp:89) */
ndtgpu_status ndtgpu_mapset_set_centre(ndtgpu_mapset *set, size_t map, const double centre[3]);
ndtgpu_status ndtgpu_mapset_info(const ndtgpu_mapset *set, size_t *n_maps, int32_t cells_per_axis[3],
                                 uint32_t *max_cells);

/* NDTMap::loadPointCloud(cloud, range) + computeNDTCells(CELL_UPDATE_MODE_SAMPLE_VARIANCE)
 * (fuser_hmt.cpp:225-227; ndt_odom_debug.cpp:178-179) for maps [first, first+count).
 * xyz_dev: DEVICE pointer; map k reads n_points records from
 *   (char*)xyz_dev + k*map_stride_bytes, records stride_bytes apart.
 * range_limit <= 0 disables the range filter; range_origins (HOST, 3 doubles per map, may be
 * NULL = sensor at the frame origin) gives the origin the range is measured from
 * (loadPointCloudCentroid, fuser_hmt.cpp:201-202).  Replaces the maps' content.
 * Asynchronous on `stream`. */
ndtgpu_status ndtgpu_mapset_build(ndtgpu_mapset *set, size_t first, size_t count, const void *xyz_dev,
                                  size_t n_points, size_t stride_bytes, size_t map_stride_bytes,
                                  double range_limit, const double *range_origins,
                                  const ndtgpu_cell_params *cell, ndtgpu_stream stream);
/* same, points in HOST memory (one H2D copy; the reference hands over host PointClouds) */
ndtgpu_status ndtgpu_mapset_build_host(ndtgpu_mapset *set, size_t first, size_t count, const void *xyz_host,
                                       size_t n_points, size_t stride_bytes, size_t map_stride_bytes,
                                       double range_limit, const double *range_origins,
                                       const ndtgpu_cell_params *cell);
/* The same with an explicit stream: returns as soon as the caller's memory has been read (it may be reused); the maps are
 * complete when `stream` is.  Both forms cut batches of >= 24 MB into chunks of clouds that travel through a ring of
 * pinned slots filled by host threads: the copy of chunk k + 1 runs under the build of chunk k, and nothing waits for the
 * whole device (the synchronous form waits for a stream of its own). */
ndtgpu_status ndtgpu_mapset_build_host_async(ndtgpu_mapset *set, size_t first, size_t count, const void *xyz_host,
                                             size_t n_points, size_t stride_bytes, size_t map_stride_bytes,
                                             double range_limit, const double *range_origins,
                                             const ndtgpu_cell_params *cell, ndtgpu_stream stream);

/* NDTMap::numberOfActiveCells / getAllCells (fuser_hmt.cpp:234; ndtgraph_conversion.h:34-43):
 * Gaussian cells in slot order (x-major, y, z).  Synchronises the build stream.
 * Any output pointer may be NULL.  cov9 row-major 3x3, idx3 = LazyGrid cell indices. */
ndtgpu_status ndtgpu_mapset_num_cells(ndtgpu_mapset *set, size_t map, uint32_t *n);
ndtgpu_status ndtgpu_mapset_export_cells(ndtgpu_mapset *set, size_t map, double *mean3, double *cov9,
                                         int32_t *idx3, uint32_t *npts);
/* installs ready-made Gaussians (CellVector::addNDTCell / a node map received from elsewhere;
 * ndt_odom_debug.cpp:217-229); the cell index is LazyGrid::getIndexForPoint(mean). */
ndtgpu_status ndtgpu_mapset_set_cells(ndtgpu_mapset *set, size_t map, const double *mean3, const double *cov9,
                                      size_t n_cells);

/* ndt_feature::discardCell(map, pt) (utils.h:229-236; ndt_feature_fuser_hmt.cpp:229-232, ndt_odom_debug.cpp:194-198): the cells that
 * hold the given points (HOST, n x packed float xyz) lose their Gaussian (hasGaussian_ = false).  Synchronous. */
ndtgpu_status ndtgpu_mapset_discard_cells(ndtgpu_mapset *set, size_t map, const float *xyz, size_t n_points);

/* ---- incremental (fused) node maps ------------------------------------------------------------- */
/* NDTMap::initialize(cx,cy,cz,sx,sy,sz) on every map of the set (fuser_hmt.cpp:89): every cell of the grid exists
 * and carries an occupancy (log-odds, 0 = no reading).  Allocates the per-slot occupancy arrays and the second cell
 * array an incremental update needs (12 bytes per slot + 80 bytes per cell and map).  Idempotent.  Plain builds
 * (ndtgpu_mapset_build*) on such a set also leave the occupancies NDTCell::computeGaussian would. */
ndtgpu_status ndtgpu_mapset_enable_occupancy(ndtgpu_mapset *set);

/* NDTMap::addPointCloud(origin, cloud, classifierTh, maxz, sensor_noise, occupancy_limit) immediately followed by
 * NDTMap::computeNDTCells(CELL_UPDATE_MODE_SAMPLE_VARIANCE, maxnumpoints, occupancy_limit, origin, noise): the two
 * calls the reference always makes together (fuser_hmt.cpp:92-94: (.., 0.1, 100.0, 0.1) + (.., 1e5, 255, ..);
 * :485-486: (.., 0.06, 25) + (.., 1e5, 255, ..)).  classifierTh, and origin / noise of computeNDTCells, are unused
 * upstream on this path and have no counterpart here. */
typedef struct {
    double maxz;            /* addPointCloud: points above it are dropped (100.0 / 25) */
    double sensor_noise;    /* addPointCloud: 0.1 / 0.06 */
    double maxnumpoints;    /* computeNDTCells: N saturates here ("sliding average"; 1e5); <= 0: never */
    double occupancy_limit; /* both: occupancy is clamped to +-limit (255) */
    int32_t n_min;          /* see ndtgpu_cell_params */
    double eval_factor;
} ndtgpu_fuse_params;
void ndtgpu_default_fuse_params(ndtgpu_fuse_params *p);   /* the values of fuser_hmt.cpp:485-486 */
/* Maps [first, first+count) each receive one cloud (map k: n_points records at xyz + k*map_stride_bytes, in the map's
 * frame) taken from sensor position origins[3*k..] (HOST).  Per beam the cells between sensor and hit receive
 * emptiness evidence, the hit joins its cell; cells merge the new points into their Gaussian (N, mean, covariance),
 * Gaussians whose occupancy falls to <= 0 disappear.  Needs ndtgpu_mapset_enable_occupancy.  xyz_dev: DEVICE pointer;
 * asynchronous on `stream`. */
ndtgpu_status ndtgpu_mapset_add_cloud(ndtgpu_mapset *set, size_t first, size_t count, const void *xyz_dev,
                                      size_t n_points, size_t stride_bytes, size_t map_stride_bytes,
                                      const double *origins, const ndtgpu_fuse_params *prm, ndtgpu_stream stream);
/* same, points in HOST memory; synchronous */
ndtgpu_status ndtgpu_mapset_add_cloud_host(ndtgpu_mapset *set, size_t first, size_t count, const void *xyz_host,
                                           size_t n_points, size_t stride_bytes, size_t map_stride_bytes,
                                           const double *origins, const ndtgpu_fuse_params *prm);
/* (asynchronous form of the host entry, like ndtgpu_mapset_build_host_async) */
ndtgpu_status ndtgpu_mapset_add_cloud_host_async(ndtgpu_mapset *set, size_t first, size_t count, const void *xyz_host,
                                                 size_t n_points, size_t stride_bytes, size_t map_stride_bytes,
                                                 const double *origins, const ndtgpu_fuse_params *prm,
                                                 ndtgpu_stream stream);
/* a fresh NDTMap in slots [first, first+count): no Gaussians, occupancy 0 (a new graph node, graph.cpp:87-101) */
ndtgpu_status ndtgpu_mapset_clear(ndtgpu_mapset *set, size_t first, size_t count);
/* NDTCell::getOccupancy of every cell of one map, slot order (x-major, y, z): cells_per_axis[0]*[1]*[2] floats */
ndtgpu_status ndtgpu_mapset_export_occupancy(ndtgpu_mapset *set, size_t map, float *occ_out);

/* the inverse of export_occupancy: installs NDTCell::occ of every cell of one map (a map received as a message,
 * ndtgraph_conversion.h:129-158).  Needs ndtgpu_mapset_enable_occupancy. */
ndtgpu_status ndtgpu_mapset_import_occupancy(ndtgpu_mapset *set, size_t map, const float *occ);

/* ndt_feature::overlapNDTOccupancyScore(ref, mov, T) (ndt_feature_node.h:213-252; used at ndt_feature_graph.cpp:175,
 * 338-340) for n_links (ref, mov, T) triples.  Both sets need ndtgpu_mapset_enable_occupancy.  T16: HOST, n_links x 16
 * column-major.  score: HOST n_links doubles (1.0 when no cell pair overlaps); nb_sum (may be NULL): HOST, the number
 * of compared cell pairs.  Synchronous. */
ndtgpu_status ndtgpu_overlap_score_batch(ndtgpu_mapset *ref_set, const uint32_t *ref_idx, ndtgpu_mapset *mov_set,
                                         const uint32_t *mov_idx, const double *T16, size_t n_links, double *score,
                                         int64_t *nb_sum, ndtgpu_stream stream);

/* ---- matcher ---------------------------------------------------------------------------- */
/* NDTMatcherD2D::derivativesNDT(sourceCells, targetMap, g, H, computeHessian)
 * (ndt_matcher_d2d_fusion.h:80, 238, 444, 617, 856, 1085): lets the in-repo matchFusion host
 * loop run unchanged.  src_* are HOST arrays of m cells already in the target frame.
 * g[6]; H[36] row-major (untouched when compute_hessian == 0). */
ndtgpu_status ndtgpu_derivatives(ndtgpu_mapset *target, size_t target_map, const double *src_mean3,
                                 const double *src_cov9, size_t m, int n_neighbours, int compute_hessian,
                                 double lfd1, double lfd2, double *score, double g[6], double H[36]);

/* NDTMatcherD2D::match(target, source, T, useInitialGuess) (ndt_feature_graph.cpp:273) and
 * NDTMatcherD2D_2D::match (ndt_matcher_d2d_fusion.h:1175) for n_pairs independent pairs --
 * the loop of NDTFeatureGraph::updateLinksUsingNDTRegistration (ndt_feature_graph.cpp:347-353).
 * Pair k matches target_set[target_idx[k]] (fixed) against source_set[source_idx[k]] (moving).
 * T16: HOST, n_pairs x 16 doubles, in: initial guess, out: result.  results: HOST, n_pairs.
 * The whole Newton / More-Thuente loop runs on the device: persistent workgroups pulling pairs from a ticket
 * counter when the batch fills the chip, one grid-barrier launch with several workgroups per registration when it
 * does not (<= 8 pairs, or <= 128 pairs of large maps; the one-link-at-a-time call of ndt_feature_graph.cpp:273 is
 * n_pairs = 1).
 * Synchronous (returns after the results are on the host). */
ndtgpu_status ndtgpu_match_batch(ndtgpu_mapset *target_set, const uint32_t *target_idx, ndtgpu_mapset *source_set,
                                 const uint32_t *source_idx, double *T16, size_t n_pairs,
                                 const ndtgpu_match_params *prm, ndtgpu_match_result *results,
                                 ndtgpu_stream stream);
/* device-resident variant for pipelines: T16_dev / results_dev are DEVICE buffers, idx arrays
 * DEVICE uint32.  ONE launch, ALWAYS asynchronous on `stream`: no host synchronisation.
 * Batches that fill the chip (and every batch on sets of small maps): persistent workgroups that pull pairs from a
 * ticket counter; safe under stream capture.  At most half as many pairs as CUs on a source set with room for >= 16384
 * cells per map (3D maps): several workgroups per registration -- static teams at a grid barrier up to 8 pairs, beyond
 * that a pool of (registration, evaluation, chunk) tasks that any workgroup serves (32 pairs of 12 k-cell maps: 7 instead
 * of 65 ms) -- ordered behind the previous launch of its kind, on whatever stream, by an event (not while `stream` is
 * being captured: the persistent kernel then).  Environment
 * NDTGPU_DEVICE_COOP=0 (read per call) keeps every batch on the persistent kernel.  Either way a registration's result
 * does not depend on its batch; the shapes agree to 1e-8 (another summation order).  A launch that gives up (a grid
 * barrier starved by a foreign process for seconds) reports exit_code -4, converged = 0 and leaves the pose untouched.
 * The indices are range-checked on the device and a map whose build overflowed max_cells is refused: such a pair gets
 * converged = 0 and exit_code -2 / -3, its pose stays untouched.  The work area (ticket counters, parked solver
 * states) belongs to the TARGET set: calls on different streams with the same target set -- through this entry or the
 * host-pointer ones -- are ordered by an event; builds of the involved maps must be ordered before the call by the
 * caller (same stream, or an event). */
ndtgpu_status ndtgpu_match_batch_device(ndtgpu_mapset *target_set, const uint32_t *target_idx_dev,
                                        ndtgpu_mapset *source_set, const uint32_t *source_idx_dev,
                                        double *T16_dev, size_t n_pairs, const ndtgpu_match_params *prm,
                                        ndtgpu_match_result *results_dev, ndtgpu_stream stream);
/* The persistent kernel's safety valve: a wave that finds no work for ~2 s raises a word in the target set's work area
 * and every workgroup leaves; registrations that were never drawn then keep whatever results_dev held.  The
 * host-pointer entries read the word themselves (NDTGPU_ERR_HIP); after a device batch the caller asks here (waits for
 * the set's last stream).  No counterpart in the reference: graph.cpp:347-353 is a serial host loop. */
ndtgpu_status ndtgpu_match_aborted(ndtgpu_mapset *target_set, int *aborted);
/* ndt_feature::matchFusion(target, source, <empty feature maps>, T, Tcov, useInitialGuess, useNDT = true,
 * useFeat = false, step_control, ITR_MAX, n_neighbours, DELTA_SCORE, useSoftConstraints, ...,
 * useTikhonovRegularization = false)  (ndt_matcher_d2d_fusion.h:797-1155; call site
 * ndt_feature_fuser_hmt.cpp:356-357): the D2D matcher plus the odometry soft constraint
 * x^T Tcov^-1 x on the accumulated pose increment (fusion.h:875-890, 1098-1110).
 * Tcov36: HOST, n_pairs x 36 doubles, row-major 6x6 covariance of (x,y,z,roll,pitch,yaw).
 * use_soft_constraints is a bit set: bit 0 = useSoftConstraints, bit 1 = useTikhonovRegularization (fusion.h:894-911,
 * 1113-1115: g <- H^T g + Q x0, H <- H^T H + Q, score += x0^T Q x0 with x0 the 2D pose vector of T Tinit^-1 and
 * Q = Tcov^-1); 0 degenerates to ndtgpu_match_batch.  Feature / odometry-cell maps: ndtgpu_match_fusion_feat_batch. */
ndtgpu_status ndtgpu_match_fusion_batch(ndtgpu_mapset *target_set, const uint32_t *target_idx,
                                        ndtgpu_mapset *source_set, const uint32_t *source_idx, double *T16,
                                        const double *Tcov36, size_t n_pairs, const ndtgpu_match_params *prm,
                                        int use_soft_constraints, ndtgpu_match_result *results, ndtgpu_stream stream);
/* The feature / odometry-cell maps of ndt_feature::matchFusion (targetNDT_feat, sourceNDT_feat, corr_feat;
 * ndt_matcher_d2d_fusion.h:797-801): two CellVector maps whose cells correspond one to one.  The fuser fills them with
 * FLIRT matches and / or 40 copies of an odometry cell pair (ndt_feature_fuser_hmt.cpp:291-334: target mean
 * Tnow * Tmotion.translation(), source mean Tinit * 0, covariance odom_cov, the last source cell un-rotated).
 * Correspondence i of registration k is entry offsets[k] + i of the four arrays (corr_feat[i] = (i, i)); at most 64 per
 * registration.  All arrays HOST. */
typedef struct ndtgpu_feat_pairs {
    const uint32_t *offsets;     /* [n_pairs + 1], non-decreasing */
    const double *src_mean;      /* [total][3]   cells of sourceNDT_feat: moved by T like the source map */
    const double *src_cov;       /* [total][6]   xx xy xz yy yz zz */
    const double *tgt_mean;      /* [total][3]   cells of targetNDT_feat */
    const double *tgt_cov;       /* [total][6] */
} ndtgpu_feat_pairs;
/* ndt_feature::matchFusion with useFeat (ndt_matcher_d2d_fusion.h:797-1155; call site ndt_feature_fuser_hmt.cpp:353-357
 * with use_odom_or_features): per Newton iteration the sums of NDTMatcherFeatureD2D::derivativesNDT over the
 * correspondences (the same pair term as the D2D matcher, known correspondence) are added to the NDT sums (fusion.h:858-
 * 871); with step control NDTMatcherD2D::lineSearchMT runs on the NDT maps, then NDTMatcherFeatureD2D::lineSearchMT on
 * the feature maps with the (possibly negated) increment the first one left, and the step is the smaller of the two, the
 * larger one when either is 0 (fusion.h:1013-1023); the final score includes the feature score (fusion.h:1085-1096).
 * flags: bit 0 useSoftConstraints, bit 1 useTikhonovRegularization, bit 2 step_control_fusion.  With bit 2 set and bit 0
 * clear the reference runs the JOINT line search lineSearchMTFusion (fusion.h:1004-1006, 390-793) instead: one search
 * on f_ndt(trial) + f_feat -- where, as written upstream, the feature maps are evaluated on the UN-stepped cells in every
 * trial (fusion.h:619), so their score and gradient at the current pose enter as constants.  Restated as written.
 * `fevals` counts derivative evaluations of the NDT maps.  feat == NULL: ndtgpu_match_fusion_batch.
 * A registration WITHOUT correspondences (offsets[k + 1] == offsets[k]) inside such a batch is run with useFeat = false, i.e.
 * by the rules of ndtgpu_match_fusion_batch: the reference's only call site passes useFeat = true together with at least
 * one correspondence (FLIRT matches that passed the consistency check, or the 40 odometry cells:
 * ndt_feature_fuser_hmt.cpp:296-320, 341-347).  What upstream's feature line search would do on EMPTY maps -- a zero
 * directional derivative, hence the in-place negation of the increment and the recovery step -- is deliberately not
 * restated, neither here nor in oracle/ndt_oracle.c (match_common: use_feat = n_feat > 0). */
ndtgpu_status ndtgpu_match_fusion_feat_batch(ndtgpu_mapset *target_set, const uint32_t *target_idx,
                                             ndtgpu_mapset *source_set, const uint32_t *source_idx, double *T16,
                                             const double *Tcov36, const ndtgpu_feat_pairs *feat, size_t n_pairs,
                                             const ndtgpu_match_params *prm, int flags, ndtgpu_match_result *results,
                                             ndtgpu_stream stream);
/* NDTMatcherD2D::covariance(target, source, T, cov) for n_links registered links (ndt_feature_graph.cpp:296-298;
 * ndt_feature_fuser_hmt.cpp:403-405): cov = H^-1 (0.03^2 J^T J) H^-1, H = D2D Hessian at T (prm->n_neighbours, lfd1,
 * lfd2), one row of J per source cell that falls into a Gaussian target cell.  PROVENANCE: perception_oru, restated from
 * memory (the least certain part of the path); `mode` selects what the row formula uses for the
 * pose Jacobians: 0 = those of the source cell itself (computeDerivativesLocal), 1 = the matcher's constructor values
 * (j = [I 0], Z = 0), which is what revisions whose derivativesNDT works on thread-local copies effectively compute.
 * T16: HOST n_links x 16 column-major (the registered poses); cov36: HOST n_links x 36 row-major.  A singular Hessian
 * gives an all-zero matrix for that link and NDTGPU_ERR_INVALID is NOT raised: singular[k] (may be NULL) is set to 1. */
ndtgpu_status ndtgpu_covariance_batch(ndtgpu_mapset *target_set, const uint32_t *target_idx, ndtgpu_mapset *source_set,
                                      const uint32_t *source_idx, const double *T16, size_t n_links,
                                      const ndtgpu_match_params *prm, int mode, double *cov36, int32_t *singular,
                                      ndtgpu_stream stream);
/* ---- exchange records of cell maps (multi-GPU graph replay, SURVEY.md 8e phases A-B) --------------------------------
 * With the node maps of a replay built data-parallel (node k on rank k mod world), every rank needs every node map before
 * it refines its share of the links: ndt_feature_graph.cpp:273 reads nodes_[ref].map and nodes_[mov].map (the loop
 * :347-353; candidate enumeration :395-405; caller ndt_feature_graph_opt.cpp:131-160).  pack writes ONE fixed-stride
 * record per map into a DEVICE buffer -- what one all_gather then moves --, unpack installs records into maps of a set
 * with the same grid geometry (cells, rank map, counters; occupancies when both sides carry them): an unpacked map is
 * indistinguishable from the packed one (same matcher bits).  Record = ndtgpu_packed_header, cells_cap cell records
 * (80 bytes each, the first n_cells valid, in LazyGrid slot order), then cells-per-grid floats when with_occupancy
 * (NDTCell::occ of every cell, for overlapNDTOccupancyScore, ndt_feature_node.h:213-252).  A map with more cells than
 * cells_cap is cut and flagged (flags bit 0), as is one that overflowed max_cells where it was built.  Asynchronous on
 * `stream`; no counterpart in the reference, whose graph lives in one process. */
typedef struct ndtgpu_packed_header { uint32_t n_cells, flags, n_dropped, cells_cap; } ndtgpu_packed_header;
typedef struct ndtgpu_cell_record {            /* 80 bytes */
    double mean[3];
    double cov[6];                             /* xx xy xz yy yz zz */
    uint32_t n;                                /* points behind the Gaussian */
    uint32_t slot;                             /* (ix * size_y + iy) * size_z + iz */
} ndtgpu_cell_record;
size_t ndtgpu_mapset_pack_bytes(const ndtgpu_mapset *set, uint32_t cells_cap, int with_occupancy);   /* bytes per record */
ndtgpu_status ndtgpu_mapset_pack_cells_device(ndtgpu_mapset *set, size_t first, size_t count, void *buf_dev,
                                              size_t record_stride_bytes, uint32_t cells_cap, int with_occupancy,
                                              ndtgpu_stream stream);
ndtgpu_status ndtgpu_mapset_unpack_cells_device(ndtgpu_mapset *set, size_t first, size_t count, const void *buf_dev,
                                                size_t record_stride_bytes, int with_occupancy, ndtgpu_stream stream);
/* The SPARSE form of the occupancy block: {uint32 n_occ, uint32 occ_cap} + occ_cap x {uint32 slot, float occ}, the cells
 * that have a reading (occupancy != 0, i.e. not the 0.5 of "initialised, no readings", ndt_feature_node.h:213-252) in slot
 * order; flags bit 2 marks it and ndtgpu_mapset_unpack_cells_device installs either form (every other cell gets "no
 * reading").  A fused node map of the replay has readings in 2-3 % of its 80 000 slots: 22 KB instead of 320 KB per node,
 * the exchange record 50 KB instead of 371 KB.  ndtgpu_mapset_occupied_cells_max: the largest number of such cells over
 * maps [first, first + count) -- what occ_cap must hold (host result: synchronises `stream`; with several ranks take the
 * maximum over ranks: the record stride is common).  A map with more readings than occ_cap is cut and flagged like one with
 * more cells than cells_cap. */
size_t ndtgpu_mapset_pack_bytes_sparse(const ndtgpu_mapset *set, uint32_t cells_cap, uint32_t occ_cap);
ndtgpu_status ndtgpu_mapset_occupied_cells_max(ndtgpu_mapset *set, size_t first, size_t count, uint32_t *max_occupied,
                                               ndtgpu_stream stream);
ndtgpu_status ndtgpu_mapset_pack_cells_sparse_device(ndtgpu_mapset *set, size_t first, size_t count, void *buf_dev,
                                                     size_t record_stride_bytes, uint32_t cells_cap, uint32_t occ_cap,
                                                     ndtgpu_stream stream);

/* ---- scans in, poses out: the whole path as ONE call ------------------------------------------------------------------
 * The reference reaches the path through calls that do everything for their inputs at once:
 * NDTFeatureGraph::updateLinksUsingNDTRegistration (ndt_feature_graph.cpp:347-353: every link of the list) and the fuser's
 * loadPointCloud + computeNDTCells + match of ndt_feature_fuser_hmt.cpp:195-227, 353-357 (raw scan -> NDT map -> pose).
 * A registrar is that call shape for batches of scan pairs: it owns `depth` internal map sets (2 x pairs_per_batch maps of
 * one grid geometry each), one stream per map set and the index arrays, and every submitted batch travels
 *     grid build of all target and source scans of a sub-batch (ONE launch)  ->  D2D matcher (ONE launch)
 * on the next internal stream, released when the previous sub-batch's builds are done: the builds of sub-batch k + 1 run on
 * the CUs that the long registrations of sub-batch k do not occupy (the pipeline that bench.py drove by hand until round 4).
 * Results are bit for bit those of ndtgpu_mapset_build + ndtgpu_match_batch_device on the same scans. */
typedef struct ndtgpu_registrar ndtgpu_registrar;
/* What a caller may decide about a registrar -- one POD struct, defaults from
 * ndtgpu_default_registrar_params; a field left at 0 means "the library's choice".  (The NDTGPU_REG_* environment variables of
 * earlier rounds survive as EXPERIMENT switches only: they fill in fields the caller left at 0.) */
enum { NDTGPU_MATCHER_AUTO = 0,        /* stream-fed where it applies (2 <= depth <= 8, max_cells < 16384, a device with more than
                                        * one stream priority), else one launch per sub-batch */
       NDTGPU_MATCHER_PER_BATCH = 1,   /* one matcher launch per sub-batch on the sub-batch's stream, ordered by events */
       NDTGPU_MATCHER_STREAM_FED = 2   /* ONE running matcher instance serves batch after batch from a queue in device memory;
                                        * create fails with NDTGPU_ERR_INVALID where it cannot be had */ };
typedef struct {
    size_t pairs_per_batch;   /* registrations per internal sub-batch (1024 fills an MI355X: two registrations per CU in flight) */
    int32_t depth;            /* internal map sets in flight, 1 .. 16 (the stream-fed matcher serves up to 8; a batch is complete
                               * 3-5 ms after its publication and its set is busy until then: 8 keeps the builds from waiting for
                               * that, 4 costs ~5 % on the bench; memory: ndtgpu_mapset_create x 2 x pairs_per_batch maps each) */
    int32_t matcher_form;     /* NDTGPU_MATCHER_* */
    uint32_t matcher_groups;  /* CUs (workgroups) the matcher side holds while builds run beside it; 0 = MEASURED: the first
                               * sub-batch runs alone on the chip, the kernels' own clocks give the CU-time of builds and
                               * registrations, and the chip is split in that proportion */
    int32_t build_streams;    /* stream-fed form: 1 or 2 build streams that take the sub-batches in turn; 0 = 2 when depth >= 3 */
    uint32_t linger_us;       /* stream-fed form: how long a matcher instance that has worked stays when it runs dry (0 = 1000: where
                               * the builds are the slower side an instance that leaves has to be placed again among build
                               * workgroups that keep arriving; ndtgpu_registrar_sync does not wait for it) */
    int32_t recalibrate_pct;  /* measured split only: when the mean number of Gaussian cells per map over the last sub-batches
                               * differs from the figure the split was measured at by more than this many percent, the pipeline is
                               * drained once and the split measured again (a registrar that moves from halls to clutter).
                               * 0 = 25; negative = never */
    int32_t matcher_slots;    /* stream-fed form: registrations in flight per workgroup of a matcher instance, 2 (hit lists of 1024 entries
                               * per share) or 3 (512: maps of up to ~450 cells; the one-lane solver steps of a registration are then covered
                               * by two others, +8 % on the bench); 0 = chosen with the split by the measured cells per map (2 with a
                               * forced split).  The same bits either way. */
} ndtgpu_registrar_params;
void ndtgpu_default_registrar_params(ndtgpu_registrar_params *p);
typedef struct {
    int32_t matcher_form;     /* the form in use: NDTGPU_MATCHER_PER_BATCH or NDTGPU_MATCHER_STREAM_FED */
    uint32_t matcher_groups;  /* current CUs of the matcher side (0: stream-fed and not measured yet) */
    int32_t build_streams;
    int32_t calibrations;     /* how often the split has been measured */
    uint64_t submitted;       /* sub-batches so far */
    double cells_per_map;     /* mean Gaussian cells per map of the sub-batch the split was last measured on */
    int32_t matcher_slots;    /* registrations in flight per workgroup of a matcher instance */
    int32_t resident_groups;  /* workgroups of matcher instances on the device at the time of the call (stream-fed form; a 4-byte
                               * device read on the null stream) */
} ndtgpu_registrar_info;
/* grid: as ndtgpu_mapset_create (grid->max_cells applies per scan). */
ndtgpu_status ndtgpu_registrar_create_ex(const ndtgpu_grid_params *grid, const ndtgpu_registrar_params *params,
                                         ndtgpu_registrar **out);
/* ... with the default parameters but for the two that every caller has to think about */
ndtgpu_status ndtgpu_registrar_create(const ndtgpu_grid_params *grid, size_t pairs_per_batch, int depth,
                                      ndtgpu_registrar **out);
ndtgpu_status ndtgpu_registrar_get_info(const ndtgpu_registrar *reg, ndtgpu_registrar_info *info);
/* TEST AID (stream-fed form): raises the matcher's abort word, as a workgroup that found no work for ~30 s would -- the give-up
 * protocol (results "not run", map sets held until the instances have left, one error from ndtgpu_registrar_sync, a registrar
 * that works again afterwards) is otherwise only reachable by starving the device for half a minute. */
ndtgpu_status ndtgpu_registrar_inject_abort(ndtgpu_registrar *reg);
ndtgpu_status ndtgpu_registrar_destroy(ndtgpu_registrar *reg);
/* n_pairs registrations: pair k builds the target map from cloud k of targets_dev and the source map from cloud k of
 * sources_dev (DEVICE pointers; n_points records each, stride_bytes apart, clouds map_stride_bytes apart; range filter as
 * ndtgpu_mapset_build with the sensor at the frame origin) and runs NDTMatcherD2D::match(target, source, T, prm).
 * T16_dev: DEVICE, n_pairs x 16 doubles column-major, in: initial guess, out: registered pose; results_dev: DEVICE.
 * ASYNCHRONOUS: the inputs must be complete in `stream` order at the time of the call (the library records an event there);
 * the work itself runs on the registrar's own streams, n_pairs > pairs_per_batch in sub-batches on successive streams.
 * The call neither waits for the device nor makes `stream` wait: successive calls overlap (whatever streams they name).
 * *ticket (may be NULL) names the call: its outputs are complete -- and its input and output buffers may be reused -- once
 * ndtgpu_registrar_wait_stream(reg, ticket, s) has made a stream wait for it, or ndtgpu_registrar_sync has returned.  A pair
 * whose map overflowed grid->max_cells reports exit_code -3 like ndtgpu_match_batch_device. */
ndtgpu_status ndtgpu_register_batch_device(ndtgpu_registrar *reg, const void *targets_dev, const void *sources_dev,
                                           size_t n_points, size_t stride_bytes, size_t map_stride_bytes, double range_limit,
                                           const ndtgpu_cell_params *cell, double *T16_dev, size_t n_pairs,
                                           const ndtgpu_match_params *prm, ndtgpu_match_result *results_dev,
                                           ndtgpu_stream stream, uint64_t *ticket);
/* The same with the scans, the poses and the results in HOST memory -- what the reference's call sites hold (pcl::PointCloud,
 * Eigen::Affine3d).  Sub-batch after sub-batch the clouds are copied to the device under the builds and registrations of the
 * sub-batch before; clouds must not overlap (map_stride_bytes >= n_points * stride_bytes).  Synchronous: returns with T16
 * and results filled in (and everything submitted earlier through the device entry complete). */
ndtgpu_status ndtgpu_register_batch_host(ndtgpu_registrar *reg, const void *targets_host, const void *sources_host,
                                         size_t n_points, size_t stride_bytes, size_t map_stride_bytes, double range_limit,
                                         const ndtgpu_cell_params *cell, double *T16, size_t n_pairs,
                                         const ndtgpu_match_params *prm, ndtgpu_match_result *results);
/* `stream` waits (on the device, the host does not) for the call `ticket` names and every call before it; ticket 0: for
 * every call submitted so far.  Stream-fed form: the wait is a device-side kernel that ends when the running matcher instance
 * has completed the batch, so `stream` must not share the matcher stream's hardware queue -- a stream created with the HIGHEST
 * priority the device offers is refused (NDTGPU_ERR_INVALID); streams of default priority and the null stream are fine. */
ndtgpu_status ndtgpu_registrar_wait_stream(ndtgpu_registrar *reg, uint64_t ticket, ndtgpu_stream stream);
/* the host waits for every batch submitted so far; NDTGPU_ERR_HIP if a matcher launch gave up (ndtgpu_match_aborted).  In the
 * stream-fed form that is reported once: registrations of the batches that were cut short carry exit_code -4 (every result of a
 * batch is set to "not run" when the batch is published), and the registrar can be used again afterwards. */
ndtgpu_status ndtgpu_registrar_sync(ndtgpu_registrar *reg);
/* Profiling: when enabled every sub-batch's build launch is bracketed by HIP events on the internal stream it runs on, and so
 * is its matcher launch in the per-batch form.  In the stream-fed form there is no launch per batch: mean_ms[1] is then what
 * the queue saw of the batch -- from its publication (maps built) until its last registration finished, on the device's 100 MHz
 * clock -- for the last 64 sub-batches.  ndtgpu_registrar_kernel_ms waits for the recorded sub-batches (stream-fed: for
 * everything submitted), returns their number and the mean durations (mean_ms[0] build, mean_ms[1] matcher side; under a
 * pipeline these include the time the work shares the chip with its neighbours), and forgets them. */
ndtgpu_status ndtgpu_registrar_profiling(ndtgpu_registrar *reg, int on);
ndtgpu_status ndtgpu_registrar_kernel_ms(ndtgpu_registrar *reg, float mean_ms[2], int32_t *launches);
/* the internal map set a sub-batch slot uses (tests: cell-by-cell parity of what the registrar built; slot < depth).  Owned by
 * the registrar. */
ndtgpu_status ndtgpu_registrar_mapset(ndtgpu_registrar *reg, int slot, ndtgpu_mapset **set);

/* ---- the fuser's call as ONE entry: NDTFeatureFuserHMT::update for a batch of independent fusers ---------------------------
 * NDTFeatureFuserHMT::update (ndt_feature/src/ndt_feature_src/ndt_feature_fuser_hmt.cpp:108-512) is, per scan: move the scan
 * into the node map's frame (:186-190), build its NDT map on the node map's lattice (:201-227), matchFusion against the node
 * map with the odometry soft constraint / Tikhonov term / odometry cells (:291-357), the matcher's covariance (:399-413), the
 * consistency gate and the pose update (:415-474), move the raw scan by the new pose and ray-trace it into the node map
 * (:479-486) -- three host-synchronous C-ABI calls (build, match, add_cloud) with host arithmetic in between.  A fuser bank
 * holds B independent fusers (robots, bags, the node fusers of a graph: they share nothing) and does all of it for a range of
 * its slots in ONE asynchronous call: one upload of what the host derives from the odometry increments, then device work only --
 * the pose a registration yields goes into the fuse-in without visiting the host.  Results are bit for bit those of the calls
 * it replaces on the same inputs (tests/test_gpu_fuser_bank.py).
 * Covers the production configuration of the fuser -- globalTransf and loadCentroid (their defaults; the launch files never
 * change them), beHMT / visualisation / the FLIRT feature map outside the path. */
typedef struct ndtgpu_fuser_bank ndtgpu_fuser_bank;
typedef struct {
    /* NDTFeatureFuserHMT::Params (ndt_feature_fuser_hmt.h:58-207): the fields the path reads, same names in snake case */
    double resolution, map_size_x, map_size_y, map_size_z, sensor_range;
    double max_translation_norm, max_rotation_norm;
    int32_t check_consistency, fuse_incomplete;
    int32_t use_odom;              /* the 40 odometry cell pairs (:322-334) join the registration (matchFusion's useFeat) */
    int32_t neighbours, stepcontrol, itr_max;
    double delta_score;
    int32_t force_odom_as_est, fusion2d, all_matches_valid, use_soft_constraints, compute_cov, step_control_fusion, use_tikhonov;
    int32_t covariance_mode;       /* `mode` of ndtgpu_covariance_batch */
    int32_t discard_cells;         /* the scan map's cells that hold the scan's first and last point lose their Gaussian (:229-232) */
    int32_t pad_;
    /* MotionModel2d::Params (motion_model.hpp:123-136) */
    double motion_Cd, motion_Ct, motion_Dd, motion_Dt, motion_Td, motion_Tt;
    double sensor_pose[16];        /* setSensorPose: column-major like every pose of this header */
    uint32_t max_cells;            /* cell capacity of the node and scan maps (ndtgpu_grid_params.max_cells) */
} ndtgpu_fuser_params;
void ndtgpu_default_fuser_params(ndtgpu_fuser_params *p);   /* Params() and MotionModel2d::Params() of the reference, sensor at the origin */
/* What the host derives from (current pose, odometry increment) before the device takes over -- a pure function, exposed so
 * that a caller (or a test) can drive the three separate calls with exactly the inputs the bank uses. */
typedef struct {
    double Tscan[16];            /* Tinit * sensor_pose: raw scan -> node map frame (:186-190) */
    double scan_centre[3];       /* loadPointCloudCentroid's grid centre (:201-202) */
    double range_origin[3];      /* the sensor position the range limit is measured from */
    double Tcov[36];             /* TmotionCov, row-major (:137-146) */
    double odom_cov[9];          /* covariance of the odometry cells (:127-130) */
    double feat_src_mean[3];     /* the 40 correspondences of ndtgpu_match_fusion_feat_batch: source cell i ... */
    double feat_tgt_mean[3];     /* ... target cell i, */
    double feat_cov_rotated[6];  /* the covariance of all of them (xx xy xz yy yz zz) */
    double feat_cov_plain[6];    /* but the LAST source cell, which keeps the un-rotated one (:336-339) */
} ndtgpu_fuser_prepared;
ndtgpu_status ndtgpu_fuser_prepare(const ndtgpu_fuser_params *prm, const double Tnow16[16], const double Tmotion16[16],
                                   const double node_centre[3], ndtgpu_fuser_prepared *out);
typedef struct {
    double Tnow[16];             /* the fuser's pose after the update: what update() returns */
    double Tmotion_est[16];      /* the registered increment */
    double spose[16];            /* Tnow * sensor_pose: the frame the raw scan was fused in at */
    ndtgpu_match_result match;
    int32_t match_ok;            /* converged, or fuseIncomplete / allMatchesValid */
    int32_t registration_failure;/* the consistency gate fired: the pose is the odometry's */
    int32_t cov_singular, pad_;
    double posecov_mean[3];      /* current_posecov */
    double posecov[9];           /* column-major 3x3 */
} ndtgpu_fuser_result;
/* node_maps: the map set whose maps [0, n_fusers) are the fusers' node maps (a graph's pool: the caller keeps ownership), or
 * NULL: the bank makes its own with prm's map sizes.  Enables occupancy on it. */
ndtgpu_status ndtgpu_fuser_bank_create(const ndtgpu_fuser_params *prm, size_t n_fusers, ndtgpu_mapset *node_maps,
                                       ndtgpu_fuser_bank **out);
ndtgpu_status ndtgpu_fuser_bank_destroy(ndtgpu_fuser_bank *bank);
/* the node maps and the scan maps of the last update (borrowed) */
ndtgpu_status ndtgpu_fuser_bank_mapsets(ndtgpu_fuser_bank *bank, ndtgpu_mapset **node_maps, ndtgpu_mapset **scan_maps);
/* NDTFeatureFuserHMT::initialize(initPos, cloud, ...) (:65-102) for slots [first, first + count): pose initPose16[k] (HOST), the
 * node map centred on it, the first cloud (DEVICE, sensor frame; n_points records `stride_bytes` apart, clouds
 * `map_stride_bytes` apart) ray-traced in.  Asynchronous on `stream`. */
ndtgpu_status ndtgpu_fuser_initialize_batch(ndtgpu_fuser_bank *bank, size_t first, size_t count, const double *initPose16,
                                            const void *xyz_dev, size_t n_points, size_t stride_bytes, size_t map_stride_bytes,
                                            ndtgpu_stream stream);
/* NDTFeatureFuserHMT::update(Tmotion, cloud, pts, updateFeatureMap, updateNDTMap) for slots [first, first + count): Tmotion16
 * HOST (count x 16), clouds DEVICE in the sensor frame.  Asynchronous on `stream`; a call waits (on the host) for the bank's
 * previous call first: it starts from the poses that one left.  Scans of unequal length: pad with NaN points. */
ndtgpu_status ndtgpu_fuser_update_batch(ndtgpu_fuser_bank *bank, size_t first, size_t count, const double *Tmotion16,
                                        const void *xyz_dev, size_t n_points, size_t stride_bytes, size_t map_stride_bytes,
                                        int update_ndt_map, ndtgpu_stream stream);
/* the same with the clouds in HOST memory (pcl::PointCloud: 16-byte records), copied to the device on a stream of the bank's own;
 * asynchronous once the caller's memory has been read */
ndtgpu_status ndtgpu_fuser_initialize_batch_host(ndtgpu_fuser_bank *bank, size_t first, size_t count, const double *initPose16,
                                                 const void *xyz_host, size_t n_points, size_t stride_bytes, size_t map_stride_bytes);
ndtgpu_status ndtgpu_fuser_update_batch_host(ndtgpu_fuser_bank *bank, size_t first, size_t count, const double *Tmotion16,
                                             const void *xyz_host, size_t n_points, size_t stride_bytes, size_t map_stride_bytes,
                                             int update_ndt_map);
/* waits for the bank's last call; Tnow16: HOST, count x 16; results (may be NULL): the records of the last update call for the
 * slots it covered, zeroes for the others */
ndtgpu_status ndtgpu_fuser_poses(ndtgpu_fuser_bank *bank, size_t first, size_t count, double *Tnow16, ndtgpu_fuser_result *results);

/* single pair convenience == graph.cpp:273 */
ndtgpu_status ndtgpu_match_d2d(ndtgpu_mapset *target_set, size_t target_map, ndtgpu_mapset *source_set,
                               size_t source_map, double T16[16], const ndtgpu_match_params *prm,
                               ndtgpu_match_result *result);

/* ---- profiling hooks ------------------------------------------------------------------ */
/* When enabled, every build launched on `set` and every match whose TARGET set is `set` is
 * bracketed by HIP events recorded on the launch stream (kernel only: table reset and copies are
 * outside the bracket).  ndtgpu_last_kernel_ms waits for the end event and returns the duration
 * of the most recent launch of kernel `which` (0 build, 1 match). */
ndtgpu_status ndtgpu_profiling_enable(ndtgpu_mapset *set, int on);
ndtgpu_status ndtgpu_last_kernel_ms(ndtgpu_mapset *set, int which, float *ms);
/* raw per-map build counters: 8 x uint32 {n_alloc, n_cells, overflow, n_dropped, shader clocks of
 * build phases A (accumulate), B (finalise), C (rank), D (clean)} */
ndtgpu_status ndtgpu_mapset_counters(ndtgpu_mapset *set, size_t map, uint32_t out[8]);
/* kernel names as they appear in rocprofv3 --kernel-trace, for bench.py / profiles/ */
const char *ndtgpu_kernel_name(int which); /* 0 build, 1 match, 2 derivatives */

#ifdef __cplusplus
}
#endif
#endif /* NDTGPU_H */
