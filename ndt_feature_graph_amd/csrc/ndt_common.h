// ndt_common.h -- shared host/device types of the MI355X NDT front-end (product code).
//
// Data layout in HBM (DESIGN.md "Layout"):
//   rankmap uint2 [n_maps][slots/32+1]  per 32 LazyGrid slots {bit per Gaussian cell, rank of the word's first one};
//                                       slot = (ix*sy + iy)*sz + iz, like dataArray[x][y][z].  Slot -> cell rank is
//                                       ndt_rank_of(): there is no dense per-slot table (round 4 removed it)
//   wtable int32 [n_maps][slots]        build-time slot -> accumulator id (all -1 between builds)
//   bitmap u32   [n_maps][slots/32]     build-time occupancy bits (all 0 between builds)
//   cells  NdtCell [n_maps][max_cells]  80-byte records, Gaussian cells only, in slot order
//   acc    NdtAcc  [n_maps][max_cells]  80-byte fixed-point (int64) moment accumulators
//                                       (build scratch; all-zero between builds)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NDT_HD __host__ __device__ __forceinline__
#define NDT_D __device__ __forceinline__

// lambda_min <= NDT_DEGENERATE_REL * lambda_max counts as "eigenvalue <= 0" in
// NDTCell::rescaleCovariance (rank-deficient sample covariance; see DESIGN.md "Deviations").
#define NDT_DEGENERATE_REL 1e-9
// log(0.6 / (1.0 - 0.6)) as the reference's double arithmetic yields it (0.6 / 0.4 is 1.4999999999999998): the
// occupancy log-odds one point adds to its cell (NDTCell::computeGaussian)
#define NDT_LOGODD_OCC 0x1.9f323ecbf9849p-2
// Eigen computeInverseAndDetWithCheck default threshold on |det(CSum)|
#define NDT_DET_EPS 1e-12

struct alignas(16) NdtCell {   // 80 B: the algorithmic per-cell record
    double mean[3];
    double cov[6];             // xx xy xz yy yz zz
    uint32_t n;                // points that built the Gaussian
    uint32_t slot;             // linear LazyGrid slot
};
static_assert(sizeof(NdtCell) == 80, "NdtCell must be 80 bytes");

// Moment accumulators of u = (p - cell_origin)/res in 64-bit FIXED POINT: a partial sum v handed to the accumulator is
// rounded once to rint(v * 2^s) and added with a 64-bit integer atomic -- exact, hence associative, order-independent
// and bit-reproducible.  The scales keep every accumulator below 2^62 (s = 62 - log2 N for |u| <= 1/2, capped at 45):
// resolution 2^-45 of a cell (squared) per partial run, two orders below what decides a rank-deficient cell
// (NDT_DEGENERATE_REL * lambda_max ~ 1e-11 cell^2).  Ten atomics and 80 bytes per record (round 2: integer-valued
// hi / lo doubles with fp64 atomics, nineteen atomics and 160 bytes).
struct alignas(16) NdtAcc {    // 80 B
    long long n;               // point count
    long long s1[3];           // sum u     * 2^s1
    long long s2[6];           // sum u u^T * 2^s2  (xx xy xz yy yz zz)
};
static_assert(sizeof(NdtAcc) == 80, "NdtAcc must be 80 bytes");
static_assert(sizeof(NdtAcc) >= sizeof(NdtCell), "the finaliser writes the cell record over its accumulator");

struct NdtGrid {               // geometry shared by all maps of a set
    double res;
    int size[3];               // cells per axis
    int slots;                 // size[0]*size[1]*size[2]
    uint32_t max_cells;
    double half[3];            // size / 2.0 per axis, the addend of LazyGrid's index formula: made on the host so that
                               // kernels find it in scalar registers (kernel arguments), not behind an int -> fp64 conversion
};

#define NDT_RANK_SEGS 32       // at most this many workgroups rank one map (ndt_build_kernel MODE 3)

struct NdtMapCounters {        // per map, device resident
    uint32_t n_alloc;          // ids handed out during accumulation (0 between builds)
    uint32_t n_cells;          // Gaussian cells after finalize
    uint32_t overflow;         // ids requested beyond max_cells
    uint32_t n_dropped;        // points dropped (NaN / range / outside grid)
    uint32_t cyc[4];           // shader clocks of build phases A, B, C, D (profiling aid)
};

struct NdtSetView {            // what kernels see of a mapset
    NdtGrid grid;
    uint32_t n_maps;
    uint2 *rankmap;            // [n_maps][rm_stride]  per 32 slots: {.x = Gaussian-cell bits, .y = rank of the word's
                               //                      first Gaussian cell (valid when .x != 0)}: what the matcher probes,
                               //                      and the only slot -> rank index (ndt_rank_of)
    int32_t *wtable;           // [n_maps][slots]      build scratch
    uint32_t *bitmap;          // [n_maps][(slots+31)/32] build scratch
    NdtCell *cells;            // [n_maps][max_cells]
    NdtAcc *acc;               // [n_maps][max_cells]
    uint32_t *acc_slot;        // [n_maps][max_cells] slot of each accumulator id
    uint32_t *rank_agg;        // [n_maps][NDT_RANK_SEGS + 2] ranking of big grids on several workgroups: per segment
                               //   (1 << 31 | Gaussian cells in it) once counted, then a segment ticket and a done ticket; zero between builds
    NdtMapCounters *counters;  // [n_maps]
    double *centres;           // [n_maps][3]
    // incremental (fused) node maps -- allocated by ndtgpu_mapset_enable_occupancy, NULL otherwise:
    float *occ;                // [n_maps][slots]      NDTCell::occ of every cell (NDTMap::initialize: all cells exist)
    long long *occ_delta;      // [n_maps][slots]      beam evidence of one addPointCloud, exact sums in units of 2^-32
                               //                      (all 0 between calls)
    unsigned char *occ_touched;// [n_maps][ceil(slots / 256)]  1: a beam left evidence in this block of 256 slots (the finalise
                               //                      pass looks at those blocks only; all 0 between calls)
    NdtCell *cells_alt;        // [n_maps][max_cells]  second cell array: an incremental update reads the old cells
                               //                      while it writes the new ranking
    uint32_t *cell_sel;        // [n_maps]             0: the map's cells are in `cells`, 1: in `cells_alt`
};

// the cell array a map currently lives in
static inline __host__ __device__ NdtCell *ndt_cells_of(const NdtSetView &s, size_t map, uint32_t sel)
{
    return (sel ? s.cells_alt : s.cells) + map * (size_t)s.grid.max_cells;
}

// Lanes of one wave that hand data to each other through LDS (queues, lists, tables): the hardware runs a wave's
// LDS operations in program order, but the COMPILER only knows about one thread -- without a fence it may hoist a
// load above the loop in which another lane stores (seen: a flag read constant-folded to its pre-loop value).
// A wavefront-scope fence costs nothing at run time and pins the order.
#ifdef __HIPCC__
static __device__ __forceinline__ void ndt_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
#endif

// rank of the Gaussian cell in `slot` (cells are ranked in slot order), -1 when the slot holds none: NDTMap::getCellAtPoint
// on a rank-map word -- one 8-byte read of a 20 KB structure instead of a 4-byte read of a 320 KB one
static inline __host__ __device__ int ndt_rank_in_word(uint2 w, unsigned bit)
{
    if (!((w.x >> bit) & 1u)) return -1;
#ifdef __HIP_DEVICE_COMPILE__
    return (int)(w.y + (unsigned)__popc(w.x & ((1u << bit) - 1u)));
#else
    return (int)(w.y + (unsigned)__builtin_popcount(w.x & ((1u << bit) - 1u)));
#endif
}
static inline __host__ __device__ int ndt_rank_of(const uint2 *rankmap, unsigned slot)
{
    return ndt_rank_in_word(rankmap[slot >> 5], slot & 31u);
}

// words per map of NdtSetView::rankmap (+1: a probe window may read one word past its first)
static inline __host__ __device__ size_t ndt_rm_stride(const NdtGrid &g) { return (size_t)((g.slots + 31) / 32) + 1; }

struct NdtMatchParamsDev {
    int n_neighbours, itr_max, step_control, dof_mask, use_initial_guess;
    int fusion_flags;          // matchFusion with a Tcov: bit 0 soft constraint, bit 1 Tikhonov regularisation
    double delta_score, lfd1, lfd2;
};

struct NdtMatchResultDev {     // mirrors ndtgpu_match_result
    int32_t converged, iterations, fevals, exit_code;
    double score;
    int32_t n_source, n_target;
    int64_t cycles_eval, cycles_solver;
    int64_t pair_terms_g, pair_terms_h;
};

// host launchers (defined next to their kernels)
hipError_t ndt_launch_build(const NdtSetView &set, size_t first, size_t count, const void *xyz_dev, size_t n_points,
                            size_t stride_bytes, size_t map_stride_bytes, double range_limit,
                            const double *range_origins_dev, int n_min, double eval_factor, int nice, hipStream_t stream);
// the batch kernel for flat grids (csrc/ndt_build_flat.hip); ndt_launch_build hands over when ndt_build_flat_ok
bool ndt_build_flat_ok(const NdtGrid &g, int nice, int sdw);
hipError_t ndt_launch_build_flat(const NdtSetView &set, size_t first, size_t count, const void *xyz_dev, size_t n_points,
                                 int sdw, size_t map_stride_bytes, double range_limit, const double *range_origins_dev,
                                 int n_min, double eval_factor, int s1_shift, int s2_shift, hipStream_t stream);
// accumulate only (phase A of the build: points -> per-cell moment accumulators), z_max: points above it are dropped
hipError_t ndt_launch_accumulate(const NdtSetView &set, size_t first, size_t count, const void *xyz_dev, size_t n_points,
                                 size_t stride_bytes, size_t map_stride_bytes, double range_limit,
                                 const double *range_origins_dev, double z_max, int nice, int *s1_shift_out, int *s2_shift_out,
                                 hipStream_t stream);
struct NdtFuseParams {         // NDTMap::addPointCloud + computeNDTCells arguments (fuser_hmt.cpp:92-94, 485-486)
    double maxz, sensor_noise, maxnumpoints, occupancy_limit, eval_factor;
    int n_min;
};
hipError_t ndt_launch_fuse(const NdtSetView &set, size_t first, size_t count, const void *xyz_dev, size_t n_points,
                           size_t stride_bytes, size_t map_stride_bytes, const double *origins_dev,
                           const NdtFuseParams &prm, int nice, hipStream_t stream);
// true when res and every cell origin centre + (k - size/2) res of the grid are fp32 numbers (fp32 cell offsets are
// then exact, csrc/ndt_build.hip)
bool ndt_grid_is_nice(const NdtGrid &g, const double centre[3]);
hipError_t ndt_launch_discard(const NdtSetView &set, size_t map, const float *xyz_dev, size_t n_pts, hipStream_t stream);
hipError_t ndt_launch_overlap(const NdtSetView &rset, const uint32_t *ridx_dev, const NdtSetView &mset,
                              const uint32_t *midx_dev, const double *T16_dev, size_t n_links, double *score_dev,
                              long long *nb_dev, hipStream_t stream);
hipError_t ndt_launch_install_cells(const NdtSetView &set, size_t map, const NdtCell *cells_dev, size_t n_cells,
                                    hipStream_t stream);
// exchange records of cell maps (csrc/ndt_pack.hip): header 16 B + cells_cap x NdtCell [+ slots x float]
//   (with_occ 2: instead {n_occ, occ_cap} + occ_cap x {slot, float}: the cells with a reading)
hipError_t ndt_launch_pack(const NdtSetView &set, size_t first, size_t count, void *buf_dev, size_t stride, unsigned cells_cap,
                           int with_occ, unsigned occ_cap, hipStream_t stream);
size_t ndt_pack_sparse_occ_bytes(unsigned occ_cap);
hipError_t ndt_launch_occ_count(const NdtSetView &set, size_t first, const uint32_t *maps_dev, size_t count, unsigned *counts_dev,
                                hipStream_t stream);
hipError_t ndt_launch_occ_list(const NdtSetView &set, const uint32_t *maps_dev, size_t count, const unsigned *offs_dev, void *pairs_dev,
                               hipStream_t stream);
hipError_t ndt_launch_overlap_lists(const NdtSetView &rset, const uint32_t *ridx_dev, const NdtSetView &mset,
                                    const uint32_t *midx_dev, const uint32_t *list_of_link_dev, const unsigned *offs_dev,
                                    const void *pairs_dev, const double *T16_dev, size_t n_links, double *score_dev,
                                    long long *nb_dev, hipStream_t stream);
hipError_t ndt_launch_unpack(const NdtSetView &set, size_t first, size_t count, const void *buf_dev, size_t stride, int with_occ,
                             hipStream_t stream);
size_t ndt_match_work_bytes(size_t n_pairs, size_t n_slots);
size_t ndt_match_abort_offset();   // offset in the work area of the word the matcher raises when it gave up
size_t ndt_match_coop_work_bytes(size_t n_groups);
size_t ndt_match_coop_ctrl_bytes();
unsigned ndt_match_coop_capacity(int n_neighbours);
hipError_t ndt_launch_match_coop(const NdtSetView &tset, const uint32_t *tidx_dev, const NdtSetView &sset,
                                 const uint32_t *sidx_dev, double *T16_dev, size_t pair_begin, size_t pair_count,
                                 const NdtMatchParamsDev &prm, NdtMatchResultDev *res_dev, const double *Q36_dev,
                                 unsigned n_groups, unsigned cells_per_group, void *work_dev, size_t work_stride, int checked,
                                 hipStream_t stream, unsigned *done_host = nullptr);
hipError_t ndt_launch_match(const NdtSetView &tset, const uint32_t *tidx_dev, const NdtSetView &sset,
                            const uint32_t *sidx_dev, double *T16_dev, size_t n_pairs, const NdtMatchParamsDev &prm,
                            NdtMatchResultDev *res_dev, const double *Q36_dev, const unsigned *feat_off_dev,
                            const double *feat_cells_dev, unsigned n_groups, int park_iters, int slots,
                            unsigned double_thresh, void *work_dev, hipStream_t stream);
hipError_t ndt_launch_covariance(const NdtSetView &tset, const uint32_t *tidx_dev, const NdtSetView &sset,
                                 const uint32_t *sidx_dev, const double *T16_dev, size_t n_links, int n_neighbours,
                                 double lfd1, double lfd2, int mode, double *cov36_dev, int *status_dev, hipStream_t stream);
// the stream-fed matcher of the registrar (csrc/ndt_match.hip): a queue of published batches in device memory
size_t ndt_stream_queue_bytes();
size_t ndt_stream_abort_offset();
size_t ndt_stream_live_offset();
unsigned ndt_stream_ring();
hipError_t ndt_stream_publish(void *queue_dev, const NdtSetView &set, double *T16_dev, NdtMatchResultDev *res_dev,
                              const NdtMatchParamsDev &prm, unsigned n_pairs, unsigned seq, hipStream_t stream);
size_t ndt_stream_ring_offset();
hipError_t ndt_stream_wait(void *queue_dev, unsigned ring, unsigned seq, hipStream_t stream);   // `stream` waits until batch `seq` is complete
hipError_t ndt_stream_skip(void *queue_dev, unsigned seq, hipStream_t stream);
hipError_t ndt_launch_match_stream(void *queue_dev, int n_neighbours, int slots, unsigned n_groups, hipStream_t stream);
unsigned ndt_stream_stamps();
hipError_t ndt_stream_final(void *queue_dev, unsigned published, hipStream_t stream);     // instances stop lingering once `published` batches are complete
hipError_t ndt_stream_reset(void *queue_dev, unsigned submitted, unsigned ring);                 // after an abort, streams idle
hipError_t ndt_stream_read_stamps(const void *queue_dev, unsigned seq, unsigned long long out[2]);   // 100 MHz: published, complete
size_t ndt_match_pool_ctrl_bytes();
size_t ndt_match_pool_head_bytes();
size_t ndt_match_pool_pair_bytes(size_t n_chunks);
hipError_t ndt_launch_match_pool(const NdtSetView &tset, const uint32_t *tidx_dev, const NdtSetView &sset,
                                 const uint32_t *sidx_dev, double *T16_dev, size_t n_pairs, const NdtMatchParamsDev &prm,
                                 NdtMatchResultDev *res_dev, const double *Q36_dev, unsigned n_groups,
                                 unsigned cells_per_group, void *work_dev, size_t pair_stride, hipStream_t stream);
struct rigid;
hipError_t ndt_launch_eval(const NdtSetView &tset, size_t tmap, const NdtSetView &sset, size_t smap, const rigid &T,
                           int n_neighbours, int with_h, double lfd1, double lfd2, unsigned n_groups, double *partials_dev,
                           hipStream_t stream);
hipError_t ndt_launch_derivatives(const NdtSetView &tset, size_t tmap, const NdtCell *src_cells_dev, size_t m,
                                  int n_neighbours, int compute_hessian, double lfd1, double lfd2, double *out28_dev,
                                  hipStream_t stream);
