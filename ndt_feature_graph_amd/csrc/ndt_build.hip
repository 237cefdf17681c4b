// ndt_build.hip -- NDT grid build on CDNA4 (gfx950): K1 voxel keying, K2 per-cell moment
// accumulation, K3 Gaussian finalisation, fused in ONE kernel with one workgroup per map.
//
// Replaces (reference call sites, perception_oru semantics per SURVEY.md App. A):
//   LazyGrid::getIndexForPoint / addPoint, NDTMap::loadPointCloud(cloud, range)
//       ndt_feature/src/ndt_feature_src/ndt_feature_fuser_hmt.cpp:195-226
//   NDTMap::computeNDTCells(CELL_UPDATE_MODE_SAMPLE_VARIANCE) -> NDTCell::computeGaussian +
//   rescaleCovariance                                      ...fuser_hmt.cpp:227, ndt_odom_debug.cpp:179
//
// Design:
//   * batches (>= 256 maps): one 256-thread workgroup owns one map (MODE 0) -- its tables, accumulators and cells
//     are touched by this workgroup only; 1024 scans = 1024 workgroups, 2 waves per SIMD (register bound).
//     Few maps: MODE 1 spreads the accumulation of one scan over many workgroups (agent-scope atomics), MODE 2
//     finalises on up to 32 workgroups per map.
//   * phase A streams the raw scan ONCE.  Each wave owns a contiguous share of the scan and walks it in
//     super-tiles of 8 rounds x 512 points; a round is pulled with fully coalesced dword loads into a padded LDS
//     tile, every lane then walks its own 8 CONSECUTIVE points (an angularly ordered sweep keeps them in one cell,
//     range noise on a wall that hugs a cell face flips them between two), accumulating count, sum d and
//     sum d d^T of d = p - cell_origin (|d| <= one cell: no cancellation) in fp64 for TWO cells at once, both
//     in registers; a third cell replaces the least recently used run, which is added to a small per-wave LDS
//     table keyed by cell.  At the end of the super-tile contiguous lanes that hold the same cell are merged by a
//     segmented wavefront scan and the segment heads (and the table entries) append one record each to a flush
//     list that is drained with wide atomics.
//   * the partial sums are split into integer-valued hi/lo doubles (resolution 2^-(s+32)) before they
//     are added, with scales that keep every accumulator below 2^53: global_atomic_add_f64 is then
//     exact, hence associative -- the result does not depend on the order in which waves reach
//     the atomics.
//   * phase B: moments -> mean / sample covariance, 3x3 register-resident Jacobi, eigenvalue floor
//     (NDTCell::rescaleCovariance).
//   * phase C ranks the Gaussian cells in slot order from an occupancy BITMAP (slots/32 words, a few
//     KB) instead of scanning the dense table, writes the 80-byte cell records and the slot -> rank
//     table the matcher probes, and restores every scratch structure (work table, bitmap,
//     accumulators) to its clean state: no memset is ever issued between builds.
//   HBM algorithmic bytes per scan: 12*N (points) + 80*M (cell records)  (SURVEY.md 8d).
#include "ndt_math.h"
#include "ndt_binning.h"
#include "ndt_wave.h"
#include <cstdlib>

#define NDT_BUILD_THREADS 256
#define NDT_BUILD_WAVES (NDT_BUILD_THREADS / 64)
#define NDT_FIN_THREADS 1024  // finalise-only launch (MODE 2): more waves to hide the dependent table loads
#ifndef NDT_PPL
#define NDT_PPL 8            // consecutive points per lane per tile
#endif
#define NDT_TILE (64 * NDT_PPL)
#ifndef NDT_ROUNDS
#define NDT_ROUNDS 8
#endif
//        // sub-tiles per super-tile (one wavefront merge + flush per 2048 points)
#define NDT_IDC 64           // entries of the per-wave slot -> id cache
#define NDT_FLCAP 80         // records in the per-wave flush list (it reuses the tile buffer: 64*25*4 B / 80 B)
#define NDT_QRUNS 16         // per-wave table of replaced runs, keyed by cell (power of two)
#define NDT_EMPTY (-1)

namespace {

constexpr int ndt_gcd(int a, int b) { return b == 0 ? a : ndt_gcd(b, a % b); }

struct BuildCtx {
    int32_t *wtable;      // work table: slot -> accumulator id while building, EMPTY otherwise
    uint32_t *bitmap;     // occupancy bit per slot (set when an id is assigned)
    NdtAcc *acc;
    uint32_t *acc_slot;
    NdtMapCounters *ctr;
    uint32_t cap;
    double q1, q2;        // 2^s1 / res, 2^s2 / res^2: metres -> scaled cell units
    int dbg;
    unsigned long long *idc;   // per-wave LDS cache (slot << 32 | id), direct mapped, NDT_IDC entries
};

// one 64-bit LDS atomic: lanes that map to the same entry in the same instruction cannot tear it
NDT_D void idc_put(unsigned long long *ce, int slot, int id)
{
    __hip_atomic_exchange(ce, ((unsigned long long)(unsigned)slot << 32) | (unsigned)id, __ATOMIC_RELAXED,
                          __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// slot -> accumulator id, allocating on first touch.  Lock-free: a racing loser wastes one id
// (left with n == 0, skipped by the finaliser).  Agent scope: with MODE 1 the workgroups of one map run on
// different XCDs, whose L2s only agree on agent-scope atomics.
NDT_D int get_or_assign(const BuildCtx &b, int slot)
{
    // ids never change once assigned, so a per-wave LDS cache can answer without the global round trip
    // that would otherwise stall the wave at every flush (consecutive tiles revisit the same cells)
    unsigned long long *ce = b.idc + (slot & (NDT_IDC - 1));
    unsigned long long c = *ce;
    if ((int)(c >> 32) == slot) return (int)(unsigned)c;
    int id = b.wtable[slot];   // may be a stale EMPTY from L1; a non-EMPTY value is always final
    // ... so an EMPTY is confirmed at the memory side before an id is drawn for the slot: an id drawn for a slot that
    // already has one is lost for good (n_alloc only grows), and split launches of unordered clouds lost enough of
    // them that way to run a map out of accumulators that would have fitted
    if (id == NDT_EMPTY) id = __hip_atomic_load(&b.wtable[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (id != NDT_EMPTY) { idc_put(ce, slot, id); return id; }
    int expected = NDT_EMPTY;
    unsigned nid = __hip_atomic_fetch_add(&b.ctr->n_alloc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_compare_exchange_strong(&b.wtable[slot], &expected, (int)nid, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_fetch_or(&b.bitmap[slot >> 5], 1u << (slot & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (nid < b.cap) b.acc_slot[nid] = (uint32_t)slot;
        else __hip_atomic_store(&b.ctr->overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        idc_put(ce, slot, (int)nid);
        return (int)nid;
    }
    idc_put(ce, slot, expected);
    return expected;   // somebody else assigned it first
}

// slot -> accumulator id for up to 64 records at once (lane = record): what a 3D sweep needs, whose waves meet ~640
// distinct cells per super-tile.  One L2-served load for all the lanes, ONE addition to the map's allocation counter for
// all the new cells of the wave (the lanes take consecutive ids), one compare-and-swap each: three dependent round trips per
// 64 records, where a look-up per replaced run was up to four per POINT step of the wave.
NDT_D int get_or_assign_wave(const BuildCtx &b, int slot, bool active, unsigned lane)
{
    unsigned long long *ce = b.idc + (slot & (NDT_IDC - 1));
    int id = NDT_EMPTY;
    bool hit = false;
    if (active) {
        const unsigned long long c = *ce;
        if ((int)(c >> 32) == slot) { id = (int)(unsigned)c; hit = true; }
    }
    const bool miss = active && !hit;
    // (a memory-side atomic drops the line from the L2 it passes; another XCD's L2 may still hold an EMPTY: the
    // compare-and-swap below then fails and returns the id, at the price of the id drawn for nothing)
    if (miss) id = __hip_atomic_load(&b.wtable[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool need = miss && id == NDT_EMPTY;
    const unsigned long long nm = ndt_ballot(need);
    if (nm) {
        const int leader = __ffsll((long long)nm) - 1;
        unsigned base = 0;
        if ((int)lane == leader)
            base = __hip_atomic_fetch_add(&b.ctr->n_alloc, (unsigned)__popcll(nm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        base = (unsigned)__shfl((int)base, leader, 64);
        if (need) {
            const unsigned long long below = lane ? (nm & (~0ull >> (64u - lane))) : 0ull;
            const unsigned nid = base + (unsigned)__popcll(below);
            int expected = NDT_EMPTY;
            if (__hip_atomic_compare_exchange_strong(&b.wtable[slot], &expected, (int)nid, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_fetch_or(&b.bitmap[slot >> 5], 1u << (slot & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (nid < b.cap) b.acc_slot[nid] = (uint32_t)slot;
                else __hip_atomic_store(&b.ctr->overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                id = (int)nid;
            } else {
                id = expected;   // somebody else assigned it first
            }
        }
    }
    if (miss) idc_put(ce, slot, id);
    return id;
}

// one moment word of a cell's accumulator += v, at the memory side, where the L2s of all XCDs agree.  (Scopes below
// "agent" compile to the same instruction on gfx950: where an atomic is performed is decided by the memory type, not by
// the instruction -- a map dealt to one XCD gains nothing.)
NDT_D void acc_add(const BuildCtx &b, unsigned long long *p, unsigned long long v)
{
#ifdef NDT_BUILD_PROF
    if (b.dbg & 4) return;   // timing experiment: no accumulator traffic
    if (b.dbg & 8) {         // timing experiment: records on a 128-byte stride (one line each; the sums land in the wrong places):
                             // 0.42 ms against 0.38 -- the atomics are not priced per line
        const size_t off = (size_t)(reinterpret_cast<char *>(p) - reinterpret_cast<char *>(b.acc));
        p = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(b.acc) + (off / 80u) * 128u + off % 80u);
    }
#endif
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One partial run {n, sum d (m), sum d d^T (m^2)} becomes a 10-word record in the per-wave flush list: the sums are
// scaled to cell units * 2^s and rounded to 64-bit integers (see NdtAcc), so that the integer atomic adds that consume
// the list are exact, hence order-independent.
template <bool DEFER = false>
NDT_D void write_flush_record(const BuildCtx &b, long long *rec, int *rec_id, int slot, double n, const double *sd,
                              const double *sdd)
{
    if (DEFER) {
        *rec_id = slot;                     // the ids of a whole list are looked up together when it is drained
    } else {
        int id = get_or_assign(b, slot);
        *rec_id = (id >= 0 && (uint32_t)id < b.cap) ? id : -1;
    }
    if (DEFER) {
        // ... and so are the fixed-point words: the list holds the RAW sums (as doubles) and the drain converts them, 64
        // (record, word) items per instruction at full width -- here the nine conversions ran for the one lane in six
        // that replaces a run in a point step (90 of the ~500 instructions of a 3D point step)
        rec[0] = __builtin_bit_cast(long long, n);
#pragma unroll
        for (int k = 0; k < 3; k++) rec[1 + k] = __builtin_bit_cast(long long, sd[k]);
#pragma unroll
        for (int k = 0; k < 6; k++) rec[4 + k] = __builtin_bit_cast(long long, sdd[k]);
        return;
    }
    rec[0] = (long long)(unsigned long long)(unsigned)n;
#pragma unroll
    for (int k = 0; k < 3; k++) rec[1 + k] = ndt_fixed_from_double(sd[k] * b.q1);
#pragma unroll
    for (int k = 0; k < 6; k++) rec[4 + k] = ndt_fixed_from_double(sdd[k] * b.q2);
}

}  // namespace

// STRIDE_DW: 3 = packed xyz, 4 = pcl::PointXYZ (16-byte records), 0 = any other stride (slow path)
// MODE 0: fused build, one workgroup per map (batches: B workgroups fill the chip)
// MODE 1: accumulate only, gridDim.x workgroups share one map (gridDim.y maps): a single scan or a small
//         batch then streams on many CUs; the atomics are memory-side, hence coherent across XCDs
// MODE 2: finalise only (phases 0, B, C, D) after a MODE 1 launch; with `dbg` != 0 phases 0 and B only
// MODE 3: phases C and D on gridDim.x workgroups per map (big grids), after a MODE 2 launch with `dbg` != 0
// (no minimum-waves hint: __launch_bounds__(256, 2) halves the speed of phase A although the register count stays at
//  250 -- measured 1.79 vs 0.94 ms -- and 3 / 4 waves per SIMD spill: 2.53 / 1.92 ms)
// SCAT: clouds whose consecutive points change cell every few points (3D sweeps): a replaced run goes straight to a
//       flush list of its own (64 records per wave, 40 KB of LDS more) instead of the small per-wave table
template <int STRIDE_DW, int MODE, bool NICE, bool SCAT>
#ifndef NDT_BUILD_WPE
#define NDT_BUILD_WPE
#endif
__global__ __launch_bounds__((MODE == 2 || MODE == 3) ? NDT_FIN_THREADS : NDT_BUILD_THREADS) NDT_BUILD_WPE void ndt_build_kernel(
    NdtSetView set, unsigned first, const char *__restrict__ xyz, unsigned n_points, unsigned stride_bytes,
    size_t map_stride_bytes, double range_limit, const double *__restrict__ range_origins, int n_min,
    double eval_factor, int s1_shift, int s2_shift, int dbg, float z_max32)
{
    constexpr int SD = STRIDE_DW ? STRIDE_DW : 3;
    constexpr int LANE_DW = NDT_PPL * SD + 1;          // +1: odd stride -> conflict-free per-lane walks
    __shared__ __attribute__((aligned(16))) float s_tile[NDT_BUILD_WAVES * 64 * (STRIDE_DW ? LANE_DW : (NDT_PPL * 3 + 1))];
    constexpr int FLC = SCAT ? 64 : NDT_FLCAP;                       // records in a wave's flush list
    __shared__ int s_flid[NDT_BUILD_WAVES * FLC];
    __shared__ long long s_list[SCAT ? NDT_BUILD_WAVES * 64 * 10 : 1];
    // (the table of replaced runs is not used by SCAT: without it three of its workgroups fit a CU's LDS, not two)
    __shared__ double s_qval[SCAT ? 1 : NDT_BUILD_WAVES * 10 * NDT_QRUNS];
    __shared__ int s_qslot[SCAT ? 1 : NDT_BUILD_WAVES * NDT_QRUNS];
    __shared__ unsigned s_qcnt[NDT_BUILD_WAVES];
    __shared__ unsigned long long s_idc[NDT_BUILD_WAVES * NDT_IDC];
    __shared__ unsigned s_wave_cnt[NDT_FIN_THREADS / 64];
    __shared__ unsigned s_base;
    __shared__ unsigned s_dropped;

    const unsigned tid = threadIdx.x;
    // the wave index through readfirstlane: everything derived from it (tile ranges, staging bases) is scalar
    const unsigned lane = tid & 63u, wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const unsigned nthreads = (MODE == 2 || MODE == 3) ? NDT_FIN_THREADS : NDT_BUILD_THREADS, nwaves = nthreads / 64;
    // MODE 1 with dbg bit 1: maps are dealt to XCDs (map m of the launch -> XCD m mod 8) and the workgroups of a map
    // are those the dispatcher places there (workgroup b of the launch runs on XCD b mod 8)
    const bool xcd_map = (MODE == 1) && (dbg & 2);
    const unsigned lin_block = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned map_local = (MODE == 0) ? blockIdx.x : xcd_map ? (lin_block & 7u) + 8u * ((lin_block >> 3) / gridDim.x) : blockIdx.y;
    // MODE 2: gridDim.x workgroups share phases 0 and B of one map; the last one to finish runs phases C and D
    const unsigned fin_parts = (MODE == 2) ? gridDim.x : 1u, fin_part = (MODE == 2) ? blockIdx.x : 0u;
    const unsigned map = first + map_local;
    const NdtGrid g = set.grid;
    const uint32_t cap = g.max_cells;
    uint2 *rankmap = set.rankmap + (size_t)map * ndt_rm_stride(g);
    const unsigned bm_words = (unsigned)((g.slots + 31) >> 5);
    NdtCell *cells = set.cells + (size_t)map * cap;          // a (re)build always lands in the first cell array
    // ... the previous content may sit in the second one (after an incremental update, csrc/ndt_fuse.hip)
    const NdtCell *cells_prev = ndt_cells_of(set, map, set.cell_sel ? set.cell_sel[map] : 0u);
    NdtMapCounters *ctr = set.counters + map;
    const double cx = set.centres[map * 3 + 0], cy = set.centres[map * 3 + 1], cz = set.centres[map * 3 + 2];
    const double res = g.res, inv_res = 1.0 / g.res;
    const double hx = g.size[0] / 2.0, hy = g.size[1] / 2.0, hz = g.size[2] / 2.0;
    BuildCtx bc;
    bc.wtable = set.wtable + (size_t)map * g.slots;
    bc.bitmap = set.bitmap + (size_t)map * bm_words;
    bc.acc = set.acc + (size_t)map * cap;
    bc.acc_slot = set.acc_slot + (size_t)map * cap;
    bc.ctr = ctr;
    bc.cap = cap;
    bc.q1 = ldexp(inv_res, s1_shift);
    bc.q2 = ldexp(inv_res * inv_res, s2_shift);
    bc.dbg = (MODE == 1) ? dbg : 0;
    bc.idc = s_idc + ((MODE == 2 || MODE == 3) ? 0u : wave) * NDT_IDC;
    if (MODE != 2 && MODE != 3) bc.idc[lane & (NDT_IDC - 1)] = ~0ull;
    double ox = 0, oy = 0, oz = 0;
    if (range_origins) { ox = range_origins[map_local * 3]; oy = range_origins[map_local * 3 + 1]; oz = range_origins[map_local * 3 + 2]; }
    const char *pts = xyz + (size_t)map_local * map_stride_bytes;
    // `nice` grids (the launcher checked it for every map of the launch): res and every cell origin c + (k - size/2) res
    // are fp32 numbers.  Then origin32 = fma(k, res32, c0) is exact, and so is the fp32 difference p - origin32 (the point
    // lies within a cell of its origin, and |origin| >= res or origin == 0: the difference needs no more bits than p has):
    // the offset of a point from its cell origin costs three fp32 operations instead of five fp64 ones per axis.
    const float res32 = (float)res;
    const float c0x32 = (float)(cx - hx * res), c0y32 = (float)(cy - hy * res), c0z32 = (float)(cz - hz * res);
    NdtBinner bn;                       // point -> cell (csrc/ndt_binning.h)
    bn.init(g, cx, cy, cz, ox, oy, oz, range_limit, z_max32);

    // ---------------- phase 0: forget the previous content of the slot -> rank table -------------
    if (MODE != 1 && MODE != 3) {
        unsigned old = ctr->n_cells;
        if (old > cap) old = cap;
        for (unsigned i = fin_part * nthreads + tid; i < old; i += nthreads * fin_parts) {
            const uint32_t sl = cells_prev[i].slot;
            rankmap[sl >> 5].x = 0u;
        }
        if (MODE == 0 && tid == 0) ctr->overflow = 0;
    }
    if (tid == 0) { s_base = 0; s_dropped = 0; }
    if (MODE != 2 && MODE != 3) {
        for (unsigned i = tid; !SCAT && i < NDT_BUILD_WAVES * NDT_QRUNS; i += nthreads) s_qslot[i] = -1;
        for (unsigned i = tid; !SCAT && i < NDT_BUILD_WAVES * 10 * NDT_QRUNS; i += nthreads) s_qval[i] = 0.0;
        if (tid < NDT_BUILD_WAVES) s_qcnt[tid] = 0u;
    }
    __syncthreads();

    // ---------------- phase A: key + accumulate ----------------------------------------------------
    long long t0 = __builtin_readcyclecounter();
    // The scan is cut into sub-tiles of 512 points (64 lanes x 8 points).  A wave owns a contiguous range of
    // sub-tiles and walks it in SUPER-TILES of up to NDT_ROUNDS sub-tiles: lane l owns 8*R consecutive points of
    // the super-tile and visits them in R rounds through the same 8-point LDS row, so the wavefront merge and
    // the flush run once per up to 2048 points while the LDS tile stays 6.4 KB.
    const unsigned n_tiles = (n_points + NDT_TILE - 1) / NDT_TILE;
    // MODE 1: this workgroup's share of the map's sub-tiles; otherwise all of them.  Waves split the share.
    const unsigned n_parts = (MODE == 1) ? gridDim.x : 1u, part = (MODE == 1) ? (xcd_map ? (lin_block >> 3) % gridDim.x : blockIdx.x) : 0u;
    const unsigned tiles_per_part = (n_tiles + n_parts - 1) / n_parts;
    const unsigned part_begin = min(n_tiles, part * tiles_per_part), part_end = min(n_tiles, part_begin + tiles_per_part);
    const unsigned tiles_per_wave = (part_end - part_begin + NDT_BUILD_WAVES - 1) / NDT_BUILD_WAVES;
    const unsigned tile_begin = (MODE == 2 || MODE == 3) ? 0u : min(part_end, part_begin + wave * tiles_per_wave);
    const unsigned tile_end = (MODE == 2 || MODE == 3) ? 0u : min(part_end, tile_begin + tiles_per_wave);
    const unsigned awave = (MODE == 2 || MODE == 3) ? 0u : wave;   // phase-A per-wave LDS regions (unused when finalising)
    // the flush list below lives in the same bytes as doubles: the float view may alias it (no type-based reordering
    // of the next round's staging stores against the list's loads)
    typedef float __attribute__((may_alias)) tile_f32;
    tile_f32 *mytile = s_tile + awave * 64 * (STRIDE_DW ? LANE_DW : (NDT_PPL * 3 + 1));
    // flush list: after the point loop the tile buffer is dead and holds the records of the partial
    // runs that must be added to their cells; ONE atomic instruction then serves up to 64 (record,
    // component) items, instead of 19 dependent single-lane atomics per run.
    long long *fl_val = SCAT ? s_list + awave * (64 * 10) : reinterpret_cast<long long *>(mytile);
    int *fl_id = s_flid + awave * FLC;
    unsigned nfl = 0;   // wave-uniform
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64u - lane));
#ifdef NDT_BUILD_PROF
    long long pa_stage = 0, pa_points = 0, pa_drain = 0, pa_end = 0;
#endif
    auto drain_list = [&]() {
#ifdef NDT_BUILD_PROF
        const long long td0 = __builtin_readcyclecounter();
#endif
        ndt_wave_sync();                                  // the records were written by other lanes
        if (SCAT) {                                       // (FLC == 64: one record per lane) slots -> ids, all at once
            const bool act = lane < nfl;
            const int id = get_or_assign_wave(bc, act ? fl_id[lane] : 0, act, lane);
            if (act) fl_id[lane] = (id >= 0 && (uint32_t)id < bc.cap) ? id : -1;
            ndt_wave_sync();
        }
        const unsigned items = nfl * 10u;
        for (unsigned it = lane; it < items; it += 64u) {
            unsigned e = it / 10u, k = it % 10u;
            int id = fl_id[e];
            long long word = fl_val[e * 10u + k];
            if (SCAT) {                                   // raw sums: word 0 the count, 1..3 sum d, 4..9 sum d d^T
                const double raw = __builtin_bit_cast(double, word);
                const long long fx = ndt_fixed_from_double(raw * (k < 4u ? bc.q1 : bc.q2));
                word = k == 0u ? (long long)(unsigned long long)(unsigned)raw : fx;
            }
            if (id >= 0) acc_add(bc, reinterpret_cast<unsigned long long *>(bc.acc + id) + k, (unsigned long long)word);
        }
        ndt_wave_sync();                                  // the list may be overwritten now
        nfl = 0;
#ifdef NDT_BUILD_PROF
        pa_drain += __builtin_readcyclecounter() - td0;
#endif
    };
    auto push_runs = [&](bool mine, int slot, double n, const double *sd3, const double *se6) {
        unsigned long long m = ndt_ballot(mine);
        while (m) {
            if (nfl == (unsigned)FLC) drain_list();
            const unsigned room = (unsigned)FLC - nfl;
            const unsigned rank = (unsigned)__popcll(m & lt_mask);
            const bool now = mine && ((m >> lane) & 1ull) && rank < room;
            if (now) write_flush_record<SCAT>(bc, fl_val + (nfl + rank) * 10u, fl_id + nfl + rank, slot, n, sd3, se6);
            const unsigned long long done = ndt_ballot(now);
            nfl += (unsigned)__popcll(done);
            m &= ~done;
        }
    };
    // evicted runs are queued in LDS and added to their cells at the end of the tile by all lanes in
    // parallel: nothing in the point loop waits for global memory
    double *q_val = s_qval + (SCAT ? 0u : awave * (10 * NDT_QRUNS));
    int *q_slot = s_qslot + (SCAT ? 0u : awave * NDT_QRUNS);
    for (unsigned tile = tile_begin; tile < tile_end;) {
        // (3D sweeps: up to 12 rounds, so that the 9 sub-tiles a wave gets of a 200 k-point sweep on 768 workgroups are ONE
        //  super-tile -- a second one of a single round paid a whole wavefront merge and flush for an eighth of the points)
        const unsigned R = min(SCAT ? 12u : (unsigned)NDT_ROUNDS, tile_end - tile);   // rounds of this super-tile
        const unsigned p0 = tile * NDT_TILE;                              // its first point
        tile += R;
        int cs0 = -2, cs1 = -3;                // empty (negative, and never equal to the "no cell" slot -1)
        int mru1 = 0;                          // 1: run 1 was used more recently than run 0
        unsigned rn = 0, rn1 = 0;              // points in the two runs (counts: one register each)
        double sd[3] = {0, 0, 0}, se[6] = {0, 0, 0, 0, 0, 0};
        double sd1[3] = {0, 0, 0}, se1[6] = {0, 0, 0, 0, 0, 0};
        auto load_point = [&](unsigned r, int j, float &fx, float &fy, float &fz) {
            if (STRIDE_DW) {
                const tile_f32 *pf = mytile + lane * LANE_DW + j * SD;
                fx = pf[0]; fy = pf[1]; fz = pf[2];
            } else {
                const unsigned i = p0 + (lane * R + r) * NDT_PPL + j;
                const bool have = i < n_points;
                const float *pf = (const float *)(pts + (size_t)(have ? i : 0u) * stride_bytes);
                fx = have ? pf[0] : __builtin_nanf(""); fy = pf[1]; fz = pf[2];
            }
        };
        // The replaced run is ADDED to the wave's LDS table entry of its cell (lanes that bounce between three or four
        // cells at a cell corner replace a run at almost every point: the table keeps one record per cell however
        // often that happens).  All additions to a wave's table come from that wave, in program order.
        auto evict_run = [&](int victim, double vn, const double *v3, const double *v6) {
            unsigned e = ((unsigned)victim * 0x9E3779B1u) >> 28;
            bool placed = false;
#pragma unroll 1
            for (int probe = 0; probe < 4 && !placed; probe++) {
                const int old = atomicCAS(&q_slot[e], -1, victim);
                if (old == -1 || old == victim) {
                    unsafeAtomicAdd(&q_val[0 * NDT_QRUNS + e], vn);
#pragma unroll
                    for (int k = 0; k < 3; k++) unsafeAtomicAdd(&q_val[(1 + k) * NDT_QRUNS + e], v3[k]);
#pragma unroll
                    for (int k = 0; k < 6; k++) unsafeAtomicAdd(&q_val[(4 + k) * NDT_QRUNS + e], v6[k]);
                    placed = true;
                } else {
                    e = (e + 1u) & (NDT_QRUNS - 1u);
                }
            }
            if (placed) {
                s_qcnt[wave] = 1u;
            } else {                             // table full (unordered cloud): add directly
                long long rec[10];
                int rid;
                write_flush_record(bc, rec, &rid, victim, vn, v3, v6);
                if (rid >= 0)
                    for (int k = 0; k < 10; k++)
                        acc_add(bc, reinterpret_cast<unsigned long long *>(bc.acc + rid) + k, (unsigned long long)rec[k]);
            }
        };
        // A lane keeps the moments of TWO cells in registers (range noise on a wall that hugs a cell face makes its
        // consecutive points alternate between two cells; a lane that walks into the next cell keeps the old one as
        // well).  A third cell replaces the run that was used least recently; the replaced run goes to the wave's LDS
        // table.  Nothing here touches LDS or memory on the common path.
        auto add_point = [&](float fx, float fy, float fz, float gx, float gy, float gz, int slot) {
            const bool newc = (slot >= 0) & (slot != cs0) & (slot != cs1);
            if (ndt_ballot(newc)) {
                if (SCAT) {
                    // every lane with a replaced run appends one record to the wave's flush list (drained with wide
                    // atomics whenever it is full)
                    const bool to1 = cs0 >= 0 && (cs1 < 0 || mru1 == 0);
                    const int victim = to1 ? cs1 : cs0;
                    // (the sums of run 0 pass through an empty asm: a select between two LOADS of the register arrays is
                    // turned into one load through a selected POINTER, and an array that is indexed that way lives in
                    // scratch memory -- both runs' second moments did, at a memory round trip per point: 112 B of scratch
                    // and 8 k clocks per point step of a wave until round 4)
                    double vn = (double)(to1 ? rn1 : rn), v3[3], v6[6];
#pragma unroll
                    for (int k = 0; k < 3; k++) { double a = sd[k]; asm volatile("" : "+v"(a)); v3[k] = to1 ? sd1[k] : a; }
#pragma unroll
                    for (int k = 0; k < 6; k++) { double a = se[k]; asm volatile("" : "+v"(a)); v6[k] = to1 ? se1[k] : a; }
                    push_runs(newc && victim >= 0, victim, vn, v3, v6);
                    if (newc) {
                        if (to1) {
                            cs1 = slot; rn1 = 0;
#pragma unroll
                            for (int k = 0; k < 3; k++) sd1[k] = 0;
#pragma unroll
                            for (int k = 0; k < 6; k++) se1[k] = 0;
                        } else {
                            cs0 = slot; rn = 0;
#pragma unroll
                            for (int k = 0; k < 3; k++) sd[k] = 0;
#pragma unroll
                            for (int k = 0; k < 6; k++) se[k] = 0;
                        }
                    }
                } else if (newc) {
                    const bool to1 = cs0 >= 0 && (cs1 < 0 || mru1 == 0);   // an empty run first, else the older one
                    // (one copy of the code per run: selecting the victim's ten sums first would cost twenty registers)
                    if (to1) {
                        if (cs1 >= 0) evict_run(cs1, (double)rn1, sd1, se1);
                        cs1 = slot; rn1 = 0;
#pragma unroll
                        for (int k = 0; k < 3; k++) sd1[k] = 0;
#pragma unroll
                        for (int k = 0; k < 6; k++) se1[k] = 0;
                    } else {
                        if (cs0 >= 0) evict_run(cs0, (double)rn, sd, se);
                        cs0 = slot; rn = 0;
#pragma unroll
                        for (int k = 0; k < 3; k++) sd[k] = 0;
#pragma unroll
                        for (int k = 0; k < 6; k++) se[k] = 0;
                    }
                }
            }
            // offset from the point's own cell origin: |d| <= one cell, no cancellation later on
            double x, y, z;
            if (NICE) {
                x = (double)(fx - fmaf(gx, res32, c0x32));
                y = (double)(fy - fmaf(gy, res32, c0y32));
                z = (double)(fz - fmaf(gz, res32, c0z32));
            } else {
                x = (double)fx - (cx + ((double)gx - hx) * res);
                y = (double)fy - (cy + ((double)gy - hy) * res);
                z = (double)fz - (cz + ((double)gz - hz) * res);
            }
            const bool in0 = slot == cs0, in1 = slot == cs1;          // cs0 / cs1 are never -1
            if (in0) {
                rn += 1u;
                sd[0] += x; sd[1] += y; sd[2] += z;
                se[0] = fma(x, x, se[0]); se[1] = fma(x, y, se[1]); se[2] = fma(x, z, se[2]);
                se[3] = fma(y, y, se[3]); se[4] = fma(y, z, se[4]); se[5] = fma(z, z, se[5]);
            }
            if (ndt_ballot(in1)) {
                if (in1) {
                    rn1 += 1u;
                    sd1[0] += x; sd1[1] += y; sd1[2] += z;
                    se1[0] = fma(x, x, se1[0]); se1[1] = fma(x, y, se1[1]); se1[2] = fma(x, z, se1[2]);
                    se1[3] = fma(y, y, se1[3]); se1[4] = fma(y, z, se1[4]); se1[5] = fma(z, z, se1[5]);
                }
            }
            mru1 = in1 ? 1 : (in0 ? 0 : mru1);
        };
#pragma unroll 1
        for (unsigned r = 0; r < R; r++) {
#ifdef NDT_BUILD_PROF
        const long long tr0 = __builtin_readcyclecounter();
#endif
        if (STRIDE_DW) {
            // Round r of the super-tile: lane `owner` needs its points [8r, 8r+8) = dwords
            // (owner*R + r)*8*SD + e, e < 8*SD, of the super-tile.  Lane l fetches item d = l + 64k
            // (owner = d / (8*SD), e = d % (8*SD)): consecutive lanes read consecutive dwords of 96-byte pieces.
            const float *src = (const float *)pts + (size_t)p0 * SD;
            const bool full = (size_t)p0 + (size_t)NDT_TILE * R <= (size_t)n_points;
            constexpr int ROW = NDT_PPL * SD;                                // dwords of one lane's 8 points
            if (full) {
                // d = lane + 64k walks (owner, e) = divmod(d, ROW) with period P in k (64 P = RS ROW): P lane constants
                // per super-tile; everything else is a wave-uniform base (scalar) or an immediate LDS offset
                constexpr int P = ROW / ndt_gcd(ROW, 64);                         // 8 points of 12 bytes: 3; of 16 bytes: 1
                constexpr int RS = 64 * P / ROW;
                static_assert((64 * P) % ROW == 0 && ROW % P == 0, "staging period");
                unsigned goff[P], loff[P];
#pragma unroll
                for (int c = 0; c < P; c++) {
                    const unsigned d = lane + 64u * c, o = d / ROW, e = d % ROW;
                    goff[c] = o * R * ROW + e;
                    loff[c] = o * LANE_DW + e;
                }
                constexpr int CH = 8;
#pragma unroll
                for (int h = 0; h < ROW / CH; h++) {
                    float tmp[CH];
#pragma unroll
                    for (int k = 0; k < CH; k++) {
                        const int kk = h * CH + k, c = kk % P, m = kk / P;
                        const float *sb = src + (size_t)((unsigned)(RS * m) * R + r) * ROW;   // wave-uniform
                        tmp[k] = sb[goff[c]];
                    }
#pragma unroll
                    for (int k = 0; k < CH; k++) {
                        const int kk = h * CH + k, c = kk % P, m = kk / P;
                        mytile[loff[c] + m * (RS * LANE_DW)] = tmp[k];
                    }
                }
            } else {
                const unsigned lim_dw = (n_points - p0) * SD;               // valid dwords from p0 on
                for (int kk = 0; kk < ROW; kk++) {
                    const unsigned d = lane + 64u * kk;
                    const unsigned g = ((d / ROW) * R + r) * ROW + d % ROW;
                    const bool have = g < lim_dw;
                    const float v = src[have ? g : 0u];
                    // past the end of the scan: NaN points, skipped below
                    mytile[(d / ROW) * LANE_DW + d % ROW] = have ? v : __builtin_nanf("");
                }
            }
        }
        ndt_wave_sync();   // a lane's row of the tile was written by other lanes
#ifdef NDT_BUILD_PROF
        const long long tr1 = __builtin_readcyclecounter();
        pa_stage += tr1 - tr0;
#endif
        // Two points per trip: their index arithmetic is independent and interleaves (the loop is latency bound at the
        // 3 waves per SIMD the register state allows); the run bookkeeping then takes them in scan order.
#pragma unroll 1
        for (int j = 0; j < NDT_PPL; j += 2) {
            float ax, ay, az, bx, by, bz;
            load_point(r, j, ax, ay, az);
            load_point(r, j + 1, bx, by, bz);
            float agx, agy, agz, bgx, bgy, bgz;
            int aslot, bslot;
            const bool na = bn.fast(ax, ay, az, agx, agy, agz, aslot);
            const bool nb = bn.fast(bx, by, bz, bgx, bgy, bgz, bslot);
            if (ndt_ballot(na | nb)) {
                if (na) bn.exact(ax, ay, az, agx, agy, agz, aslot);
                if (nb) bn.exact(bx, by, bz, bgx, bgy, bgz, bslot);
            }
            add_point(ax, ay, az, agx, agy, agz, aslot);
            add_point(bx, by, bz, bgx, bgy, bgz, bslot);
        }
#ifdef NDT_BUILD_PROF
        pa_points += __builtin_readcyclecounter() - tr1;
#endif
        }   // rounds
        {
#ifdef NDT_BUILD_PROF
            const long long te0 = __builtin_readcyclecounter();
#endif
            ndt_wave_sync();   // table / flag written by other lanes during the rounds
            // Canonical order of a lane's two runs (run 0 = smaller slot): along a wall that hugs a cell
            // face neighbouring lanes then agree on which cell is run 0 and which is run 1, so both form
            // long contiguous segments for the scans below instead of alternating lane by lane.
            const bool swap = cs1 >= 0 && cs1 < cs0;
            if (ndt_ballot(swap)) {
                if (swap) {
                    double t;
                    { const unsigned tn = rn1; rn1 = rn; rn = tn; }
#pragma unroll
                    for (int k = 0; k < 3; k++) { t = sd1[k]; sd1[k] = sd[k]; sd[k] = t; }
#pragma unroll
                    for (int k = 0; k < 6; k++) { t = se1[k]; se1[k] = se[k]; se[k] = t; }
                    int ti = cs0; cs0 = cs1; cs1 = ti;
                }
            }
#pragma unroll 1
            for (int pass = 0; pass < 2; pass++) {
                int cs = cs0;
                if (pass == 1) {
                    if (!ndt_ballot(cs1 >= 0)) break;
                    // second pass: the lanes' run 1
                    const bool has = cs1 >= 0;
                    cs = cs1;
                    rn = has ? rn1 : 0u;
#pragma unroll
                    for (int k = 0; k < 3; k++) sd[k] = has ? sd1[k] : 0.0;
#pragma unroll
                    for (int k = 0; k < 6; k++) se[k] = has ? se1[k] : 0.0;
                }
                // segmented wavefront reduction over contiguous lanes that hold the same cell
                int prev = __shfl_up(cs, 1, 64);
                bool head = (lane == 0) || (prev != cs);
                unsigned long long hm = ndt_ballot(head);
                // lanes remaining in my segment (me included): distance to the next head above me
                unsigned long long above = (lane == 63) ? 0ull : (hm >> (lane + 1));
                int rem = above ? (__ffsll((long long)above)) : (int)(64 - lane);
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    bool take = o < rem;
                    double t;
                    { const unsigned tn = __shfl_down(rn, o, 64); if (take) rn += tn; }
#pragma unroll
                    for (int k = 0; k < 3; k++) { t = __shfl_down(sd[k], o, 64); if (take) sd[k] += t; }
#pragma unroll
                    for (int k = 0; k < 6; k++) { t = __shfl_down(se[k], o, 64); if (take) se[k] += t; }
                }
                push_runs(head && cs >= 0, cs, (double)rn, sd, se);
            }
            ndt_wave_sync();
            if (!SCAT && s_qcnt[wave]) {         // records of the replaced runs; the table goes back to empty
                const unsigned ql = lane & (NDT_QRUNS - 1u);
                const int qs = q_slot[ql];
                const bool has = lane < NDT_QRUNS && qs >= 0;
                double v3[3] = {q_val[1 * NDT_QRUNS + ql], q_val[2 * NDT_QRUNS + ql], q_val[3 * NDT_QRUNS + ql]};
                double v6[6] = {q_val[4 * NDT_QRUNS + ql], q_val[5 * NDT_QRUNS + ql], q_val[6 * NDT_QRUNS + ql],
                                q_val[7 * NDT_QRUNS + ql], q_val[8 * NDT_QRUNS + ql], q_val[9 * NDT_QRUNS + ql]};
                const double vn = q_val[ql];
                if (lane < NDT_QRUNS) {
                    q_slot[ql] = -1;
#pragma unroll
                    for (int k = 0; k < 10; k++) q_val[k * NDT_QRUNS + ql] = 0.0;
                }
                if (lane == 0) s_qcnt[wave] = 0u;
                ndt_wave_sync();
                push_runs(has, qs, vn, v3, v6);
            }
            drain_list();
#ifdef NDT_BUILD_PROF
            pa_end += __builtin_readcyclecounter() - te0;
#endif
        }
    }
#ifdef NDT_BUILD_PROF
    if (MODE == 1 && wave == 0 && lane == 0) {
        atomicAdd(&ctr->cyc[0], (unsigned)(pa_stage >> 4)); atomicAdd(&ctr->cyc[1], (unsigned)(pa_points >> 4));
        atomicAdd(&ctr->cyc[2], (unsigned)(pa_drain >> 4)); atomicAdd(&ctr->cyc[3], (unsigned)(pa_end >> 4));
    }
#endif
    __syncthreads();
    if (MODE == 1) return;   // the finalise launch does the rest
    // atomics bypass the vector L1: drop lines that phase A cached before they were updated
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();

    // ---------------- phase B: moments -> Gaussian (in place in the scratch arena) -------------------
    long long t1 = __builtin_readcyclecounter();
    unsigned n_alloc = ctr->n_alloc;
    if (n_alloc > cap) n_alloc = cap;
    NdtAcc *tmp_base = bc.acc;                            // cell record written over its own accumulator
    unsigned binned = 0;
    const double IS1 = ldexp(1.0, -s1_shift), IS2 = ldexp(1.0, -s2_shift);
    for (unsigned id = fin_part * nthreads + tid; MODE != 3 && id < n_alloc; id += nthreads * fin_parts) {
        NdtAcc a = bc.acc[id];
        const unsigned slot_of_id = bc.acc_slot[id];    // (read beside the record, not after it: stale for an id without points)
        unsigned long long n = (unsigned long long)a.n;
        binned += (unsigned)n;               // points that reached a cell (the others were NaN, out of range / grid)
        if (set.occ && n > 0) {
            // NDTCell::computeGaussian on a fresh cell: occ = n log(0.6 / 0.4), clamped to the default limit 255
            // (the launcher zeroed the map's occupancies)
            const float o = (float)((double)n * NDT_LOGODD_OCC);
            set.occ[(size_t)map * g.slots + slot_of_id] = o > 255.0f ? 255.0f : o;
        }
        const unsigned slot = n ? slot_of_id : 0u;
        const int iz = slot % g.size[2], iy = (slot / g.size[2]) % g.size[1], ix = slot / (g.size[2] * g.size[1]);
        const double centre[3] = {cx + (ix - hx) * res, cy + (iy - hy) * res, cz + (iz - hz) * res};
        const NdtCell c = ndt_gaussian_from_moments(a, slot, centre, res, n_min, eval_factor, IS1, IS2);
        *reinterpret_cast<NdtCell *>(tmp_base + id) = c;
        // a touched cell without a Gaussian leaves the occupancy bitmap here, so that phase C finds exactly the
        // Gaussian cells in it (n == 0: an id wasted by an allocation race, it has no slot)
        if (n > 0 && c.n == 0) {
            const unsigned sl = slot_of_id;
            __hip_atomic_fetch_and(&bc.bitmap[sl >> 5], ~(1u << (sl & 31u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bc.wtable[sl] = NDT_EMPTY;
        }
    }
    if (binned) atomicAdd(&s_dropped, binned);
    __syncthreads();
    if (MODE == 2 && dbg != 0) {
        // big grid: the ranking is a launch of its own (MODE 3, several workgroups per map); publish this workgroup's
        // share of the binned points and stop
        if (tid == 0 && s_dropped) atomicAdd(&ctr->n_dropped, s_dropped);
        return;
    }
    if (MODE == 2 && fin_parts > 1u) {
        // publish this workgroup's share (binned points, cell records, cleared bits) and draw a ticket: the
        // workgroup that draws the last one has everybody's phase B behind it and goes on alone
        __shared__ unsigned s_ticket;
        if (tid == 0) {
            if (s_dropped) atomicAdd(&ctr->n_dropped, s_dropped);
            __threadfence();
            s_ticket = atomicAdd(&ctr->cyc[3], 1u);
        }
        __syncthreads();
        if (s_ticket != fin_parts - 1u) return;
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            s_dropped = __hip_atomic_load(&ctr->n_dropped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ctr->cyc[3] = 0u;                              // the ticket counter rests at zero between builds
        }
        __syncthreads();
    }
    // MODE 3: gridDim.x workgroups rank one map.  A workgroup takes the next SEGMENT of the bitmap (segments are handed
    // out in arrival order: whoever holds segment s only ever waits for workgroups that are already running), counts its
    // Gaussian cells, publishes the count and adds up the counts of the segments before it.
    uint32_t *agg = set.rank_agg + (size_t)map * (NDT_RANK_SEGS + 2);
    __shared__ unsigned s_seg, s_segbase;
    unsigned seg = 0, n_segs = 1;
    if (MODE == 3) {
        n_segs = gridDim.x;
        if (tid == 0) {
            s_seg = atomicAdd(&agg[NDT_RANK_SEGS], 1u);
            s_dropped = __hip_atomic_load(&ctr->n_dropped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        seg = s_seg;
    }

    // ---------------- phase C: rank Gaussian cells in slot order from the occupancy bitmap ------------
    long long t2 = __builtin_readcyclecounter();
    // Every wave owns a contiguous share of the bitmap.  Pass 1 counts its Gaussian cells, one barrier
    // turns the wave totals into bases, pass 2 assigns ranks with a running per-wave base: no barrier
    // inside the loops (a 3D grid has 200 k bitmap words).
    auto valid_bits = [&](unsigned w, unsigned bits) {
        unsigned vmask = 0;
        for (unsigned b = bits; b; b &= b - 1) {
            int bit = __ffs((int)b) - 1;
            int id = bc.wtable[w * 32 + bit];
            if (id >= 0 && (uint32_t)id < n_alloc && reinterpret_cast<const NdtCell *>(tmp_base + id)->n > 0) vmask |= 1u << bit;
        }
        return vmask;
    };
    const unsigned words_per_seg = (bm_words + n_segs - 1) / n_segs;
    const unsigned sb = min(bm_words, seg * words_per_seg), se_ = min(bm_words, sb + words_per_seg);
    const unsigned words_per_wave = (se_ - sb + nwaves - 1) / nwaves;
    const unsigned wb = min(se_, sb + wave * words_per_wave), we = min(se_, wb + words_per_wave);
    // More touched cells than accumulators (overflow): slots whose id is past the capacity still have their bit
    // and must be filtered through the work table; otherwise the bitmap IS the set of Gaussian cells.
    const bool ovf = __hip_atomic_load(&ctr->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    // phase B cleared bits with atomics (performed at the memory side): drop what this CU's L1 may still hold,
    // then plain loads, four words per lane and step so that four loads are in flight (a 3D grid has 200 k words)
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    constexpr unsigned WPL = 4;
    // The ranking launch of a big grid keeps its wave's share of the bitmap in REGISTERS when it fits (up to 8 steps of
    // 64 lanes x 4 words; 200 k words on 8 workgroups of 16 waves: 7 steps): every load of the share is in flight at once,
    // and the second pass does not read the bitmap again.  Step by step the two passes were 14 dependent round trips.
    constexpr unsigned KI = (MODE == 3) ? 8u : 1u;
    const bool cached = (MODE == 3) && !ovf && (we - wb) <= KI * 64u * WPL;
    unsigned cb[KI][WPL];
    if (cached) {
#pragma unroll
        for (unsigned it = 0; it < KI; it++)
#pragma unroll
            for (unsigned k = 0; k < WPL; k++) {
                const unsigned w = wb + (it * 64u + lane) * WPL + k;
                cb[it][k] = w < we ? bc.bitmap[w] : 0u;
            }
        unsigned cnt = 0;
#pragma unroll
        for (unsigned it = 0; it < KI; it++)
#pragma unroll
            for (unsigned k = 0; k < WPL; k++) cnt += (unsigned)__popc(cb[it][k]);
        unsigned incl = ndt_wave_incl_scan(cnt);
        if (lane == 63) s_wave_cnt[wave] = incl;
    } else {
        unsigned cnt = 0;
        for (unsigned w0 = wb + lane * WPL; w0 < we; w0 += 64u * WPL) {
            unsigned bits[WPL];
#pragma unroll
            for (unsigned k = 0; k < WPL; k++) bits[k] = (w0 + k < we) ? bc.bitmap[w0 + k] : 0u;
#pragma unroll
            for (unsigned k = 0; k < WPL; k++) cnt += (unsigned)__popc((ovf && bits[k]) ? valid_bits(w0 + k, bits[k]) : bits[k]);
        }
        unsigned incl = ndt_wave_incl_scan(cnt);
        if (lane == 63) s_wave_cnt[wave] = incl;
    }
    __syncthreads();
    unsigned running = 0, total_cells = 0;
    for (unsigned k = 0; k < nwaves; k++) {
        unsigned c2 = s_wave_cnt[k];
        if (k < wave) running += c2;
        total_cells += c2;
    }
    if (MODE == 3) {
        if (tid == 0) {
            __hip_atomic_store(&agg[seg], 0x80000000u | total_cells, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            unsigned before = 0;
            for (unsigned q = 0; q < seg; q++) {
                unsigned v;
                while (((v = __hip_atomic_load(&agg[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0x80000000u) == 0u)
                    __builtin_amdgcn_s_sleep(4);
                before += v & 0x7FFFFFFFu;
            }
            s_segbase = before;
        }
        __syncthreads();
        running += s_segbase;
        total_cells += s_segbase;      // (cells up to the end of this segment)
    }
#ifdef NDT_BUILD_PROF
    const long long tp1 = __builtin_readcyclecounter();
#endif
    if (!ovf && MODE == 3 && cached) {
        // (the share of the bitmap is in registers: see pass 1)
#pragma unroll
        for (unsigned it = 0; it < KI; it++) {
            const unsigned w0 = wb + (it * 64u + lane) * WPL;
            unsigned cnt = 0;
#pragma unroll
            for (unsigned k = 0; k < WPL; k++) cnt += (unsigned)__popc(cb[it][k]);
            if (!ndt_ballot(cnt != 0u)) continue;
            const unsigned incl = ndt_wave_incl_scan(cnt);
            unsigned before = running + incl - cnt;
            running += __shfl(incl, 63, 64);
#pragma unroll
            for (unsigned k = 0; k < WPL; k++) {
                if (cb[it][k]) {
                    rankmap[w0 + k] = make_uint2(cb[it][k], before);
                    bc.bitmap[w0 + k] = 0u;
                    before += (unsigned)__popc(cb[it][k]);
                }
            }
        }
    } else if (!ovf && MODE == 3) {
        // Big grids (round 4): the ranking launch only writes the RANK MAP -- per bitmap word its Gaussian bits and the rank
        // of its first cell -- and clears the bitmap; every cell record is then put in its place by the placement launch
        // (ndt_place_cells_kernel), one thread per accumulator id with ONE dependent load.  Until round 4 this pass also
        // wrote every cell's slot into its final record (a 4-byte store into an 80-byte record per cell) and a third pass
        // walked slot -> work table -> record -> cell array, three dependent round trips into a 1.6 GB working set:
        // 0.34 ms of the 0.88 ms that 64 sweeps took.
        for (unsigned step = wb; step < we; step += 64u * WPL) {
            const unsigned w0 = step + lane * WPL;
            unsigned bits[WPL];
#pragma unroll
            for (unsigned k = 0; k < WPL; k++) bits[k] = (w0 + k < we) ? bc.bitmap[w0 + k] : 0u;
            if (!ndt_ballot((bits[0] | bits[1] | bits[2] | bits[3]) != 0u)) continue;
            unsigned cnt = 0;
#pragma unroll
            for (unsigned k = 0; k < WPL; k++) cnt += (unsigned)__popc(bits[k]);
            const unsigned incl = ndt_wave_incl_scan(cnt);
            unsigned before = running + incl - cnt;
            running += __shfl(incl, 63, 64);
#pragma unroll
            for (unsigned k = 0; k < WPL; k++) {
                if (bits[k]) {
                    rankmap[w0 + k] = make_uint2(bits[k], before);
                    bc.bitmap[w0 + k] = 0u;
                    before += (unsigned)__popc(bits[k]);
                }
            }
        }
    } else if (!ovf) {
        // Pass 2, usual case (the bitmap holds exactly the Gaussian cells): 32 words per wave and step, one half
        // word per lane.  The lanes list their slots in LDS in slot order (rank = list position), then the whole
        // wave walks the list: work table -> record -> cell array are dependent global accesses, 64 chains at a
        // time instead of one per word (a wall of a 3D map fills whole 32-slot words).
        // Stage 1: every Gaussian slot is written into the `slot` field of its final cell record (stores only; four
        // bitmap words per lane are in flight).  Stage 2: the wave walks its rank range, 64 cells at a time:
        // slot -> work table -> record -> cell array are dependent global accesses, but 64 independent chains
        // overlap (a lane that walked the bits of its own words did them one after the other).
        const unsigned rank_begin = running;
        for (unsigned step = wb; step < we; step += 64u * WPL) {
            const unsigned w0 = step + lane * WPL;
            unsigned bits[WPL];
#pragma unroll
            for (unsigned k = 0; k < WPL; k++) bits[k] = (w0 + k < we) ? bc.bitmap[w0 + k] : 0u;
            if (!ndt_ballot((bits[0] | bits[1] | bits[2] | bits[3]) != 0u)) continue;
            unsigned cnt = 0;
#pragma unroll
            for (unsigned k = 0; k < WPL; k++) cnt += (unsigned)__popc(bits[k]);
            const unsigned incl = ndt_wave_incl_scan(cnt);
            unsigned before = running + incl - cnt;
            running += __shfl(incl, 63, 64);
#pragma unroll
            for (unsigned k = 0; k < WPL; k++) {
                if (bits[k]) {
                    rankmap[w0 + k] = make_uint2(bits[k], before);
                    bc.bitmap[w0 + k] = 0u;
                    for (unsigned b = bits[k]; b; b &= b - 1u)
                        cells[before++].slot = (w0 + k) * 32u + (unsigned)(__ffs((int)b) - 1);
                }
            }
        }
        // the slots written above are read back below by OTHER lanes of this wave.  The stores are write-through: once
        // the wave's store counter has drained they are in L2, and the agent-scope loads below read at L2, past the L1.
        // (An agent-scope release fence says the same thing but also writes the L2 back: 0.74 -> 0.94 ms per 1024
        // scans; acquire loads drop the L1 after every one of them: 0.96 ms.  Both measured.)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (unsigned r = rank_begin + lane; r < running; r += 64u) {
            const unsigned slot = __hip_atomic_load(&cells[r].slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int id = bc.wtable[slot];
            cells[r] = *reinterpret_cast<const NdtCell *>(tmp_base + id);
            bc.wtable[slot] = NDT_EMPTY;                          // work table back to its clean state
        }
    } else
    for (unsigned step = wb; step < we; step += 64u * WPL) {
        const unsigned w0 = step + lane * WPL;
        unsigned bits[WPL], vmask[WPL];
#pragma unroll
        for (unsigned k = 0; k < WPL; k++) bits[k] = (w0 + k < we) ? bc.bitmap[w0 + k] : 0u;
        if (!ndt_ballot((bits[0] | bits[1] | bits[2] | bits[3]) != 0u)) continue;
        unsigned cnt = 0;
#pragma unroll
        for (unsigned k = 0; k < WPL; k++) {
            vmask[k] = (ovf && bits[k]) ? valid_bits(w0 + k, bits[k]) : bits[k];
            cnt += (unsigned)__popc(vmask[k]);
        }
        unsigned incl = ndt_wave_incl_scan(cnt);
        unsigned before = running + incl - cnt;
        running += __shfl(incl, 63, 64);
#pragma unroll
        for (unsigned k = 0; k < WPL; k++) {
            const unsigned w = w0 + k, bk = bits[k], vk = vmask[k];
            if (vk) rankmap[w] = make_uint2(vk, before);
            for (unsigned b = bk; b; b &= b - 1) {
                int bit = __ffs((int)b) - 1;
                unsigned slot = w * 32 + bit;
                int id = bc.wtable[slot];
                if (vk & (1u << bit)) {
                    cells[before] = *reinterpret_cast<const NdtCell *>(tmp_base + id);
                    before++;
                }
                bc.wtable[slot] = NDT_EMPTY;      // work table back to its clean state
            }
            if (bk) bc.bitmap[w] = 0u;
        }
    }
    if (tid == 0) s_base = total_cells;
    __syncthreads();
#ifdef NDT_BUILD_PROF
    if (MODE == 3 && tid == 0) {
        const long long tp2 = __builtin_readcyclecounter();
        atomicAdd(&ctr->cyc[0], (unsigned)((tp1 - t0) >> 4)); atomicAdd(&ctr->cyc[1], (unsigned)((tp2 - tp1) >> 4));
    }
#endif
    if (MODE == 3) {
        // the last segment knows the number of cells; the workgroup that finishes last cleans up for everybody
        __shared__ unsigned s_done;
        if (tid == 0) {
            if (seg == n_segs - 1u) ctr->n_cells = total_cells;
            __threadfence();
            s_done = atomicAdd(&agg[NDT_RANK_SEGS + 1], 1u);
        }
        __syncthreads();
        if (s_done != n_segs - 1u) return;
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        if (tid < NDT_RANK_SEGS + 2) agg[tid] = 0u;          // the tickets and counts rest at zero between builds
    }

    // ---------------- phase D: leave the scratch zeroed, publish counters -----------------------------
    long long t3 = __builtin_readcyclecounter();
    const bool placed_later = (MODE == 3) && !ovf;         // ndt_place_cells_kernel still needs the records and n_alloc
    if (!placed_later) {
        unsigned long long *z = reinterpret_cast<unsigned long long *>(bc.acc);
        for (unsigned k = tid; k < n_alloc * 10u; k += nthreads) z[k] = 0ull;
    }
#ifdef NDT_BUILD_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#endif
    if (tid == 0) {
        if (MODE != 3) ctr->n_cells = s_base;
        if (set.cell_sel) set.cell_sel[map] = 0u;
        if (!placed_later) ctr->n_alloc = 0;
        ctr->n_dropped = n_points - s_dropped;          // s_dropped holds the number of binned points here
#ifdef NDT_BUILD_PROF
        if (MODE == 3) atomicAdd(&ctr->cyc[2], (uint32_t)((__builtin_readcyclecounter() - t3) >> 4)); else
#endif
        {
        ctr->cyc[0] = (uint32_t)(t1 - t0);
        ctr->cyc[1] = (uint32_t)(t2 - t1);
        ctr->cyc[2] = (uint32_t)(t3 - t2);
        }
        ctr->cyc[3] = 0u;                                  // reserved: ticket counter of the multi-workgroup finalise
    }
}

// After the ranking launch of a big grid (MODE 3, no overflow): every Gaussian goes from its accumulator id to its rank.
// One thread per id: the 80-byte record (coalesced), ONE dependent 8-byte load of the rank map, the record's final
// place, the work table entry back to EMPTY, the accumulator back to zero.  gridDim.x workgroups per map; the one that
// finishes last resets the map's allocation counter.
extern "C" __global__ __launch_bounds__(NDT_FIN_THREADS) void ndt_place_cells_kernel(NdtSetView set, unsigned first)
{
    const NdtGrid g = set.grid;
    const unsigned map = first + blockIdx.y, tid = threadIdx.x;
    NdtMapCounters *ctr = set.counters + map;
    if (ctr->overflow) return;                              // the ranking launch took the general path and did all of this
    const uint32_t cap = g.max_cells;
    unsigned n_alloc = ctr->n_alloc;
    if (n_alloc > cap) n_alloc = cap;
    const uint2 *rankmap = set.rankmap + (size_t)map * ndt_rm_stride(g);
    NdtCell *cells = set.cells + (size_t)map * cap;
    NdtAcc *acc = set.acc + (size_t)map * cap;
    int32_t *wtable = set.wtable + (size_t)map * g.slots;
    // A wave moves 64 records at a time.  Lane l looks at the tail of record l (n | slot, the last 8 bytes) and finds its
    // rank; then the 64 x 80 bytes are read as 320 consecutive 16-byte pieces, five per lane, and every piece goes to its
    // record's final place (five neighbouring lanes write one record) and is zeroed where it came from.
    __shared__ int s_dst[NDT_FIN_THREADS];
    const unsigned lane = tid & 63u, wave = tid >> 6;
    int *dst_of = s_dst + wave * 64u;
    for (unsigned base = (blockIdx.x * (NDT_FIN_THREADS / 64u) + wave) * 64u; base < n_alloc;
         base += NDT_FIN_THREADS * gridDim.x) {
        const unsigned nrec = min(64u, n_alloc - base), npieces = nrec * 5u;
        uint4 *src = reinterpret_cast<uint4 *>(acc + base);
        uint4 v[5];                                         // (in flight beside the tail -> rank map chain below)
#pragma unroll
        for (unsigned j = 0; j < 5u; j++) {
            const unsigned q = lane + 64u * j;
            v[j] = q < npieces ? src[q] : make_uint4(0u, 0u, 0u, 0u);
        }
        int dst = -1;
        if (lane < nrec) {
            const uint2 tail = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(acc + base + lane) + 72);
            if (tail.x > 0u) {                              // (n == 0: an id lost to an allocation race, or a cell without a Gaussian)
                const uint2 rm = rankmap[tail.y >> 5];
                dst = (int)(rm.y + (unsigned)__popc(rm.x & ((1u << (tail.y & 31u)) - 1u)));
                wtable[tail.y] = NDT_EMPTY;                 // work table back to its clean state
            }
        }
        dst_of[lane] = dst;
        ndt_wave_sync();                                    // dst_of[] was written by other lanes
#pragma unroll
        for (unsigned j = 0; j < 5u; j++) {
            const unsigned q = lane + 64u * j;
            if (q < npieces) {
                const unsigned r = q / 5u, piece = q - 5u * r;
                const int d = dst_of[r];
                if (d >= 0) reinterpret_cast<uint4 *>(cells + d)[piece] = v[j];
                src[q] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        ndt_wave_sync();                                    // dst_of[] may be overwritten now
    }
    __shared__ unsigned s_last;
    uint32_t *agg = set.rank_agg + (size_t)map * (NDT_RANK_SEGS + 2);
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&agg[NDT_RANK_SEGS + 1], 1u);
    __syncthreads();
    if (s_last == gridDim.x - 1u && tid == 0) {
        agg[NDT_RANK_SEGS + 1] = 0u;                        // the ticket rests at zero between builds
        ctr->n_alloc = 0;
    }
}

// Installs ready-made Gaussians (CellVector-like maps, KATs): cells arrive sorted by slot, one per
// slot (the host wrapper guarantees it).  The slot -> rank table was reset by the launcher.
extern "C" __global__ void ndt_install_cells_kernel(NdtSetView set, unsigned map, const NdtCell *__restrict__ src,
                                                    unsigned n_cells)
{
    const NdtGrid g = set.grid;
    uint2 *rankmap = set.rankmap + (size_t)map * ndt_rm_stride(g);
    NdtCell *cells = set.cells + (size_t)map * g.max_cells;
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_cells) {
        NdtCell c = src[i];
        cells[i] = c;
        atomicOr(&rankmap[c.slot >> 5].x, 1u << (c.slot & 31u));
        if (i == 0 || (src[i - 1].slot >> 5) != (c.slot >> 5)) rankmap[c.slot >> 5].y = i;   // sorted by slot
    }
    if (i == 0) {
        if (set.cell_sel) set.cell_sel[map] = 0u;
        set.counters[map].n_cells = n_cells;
        set.counters[map].n_alloc = 0;
        set.counters[map].overflow = 0;
        set.counters[map].n_dropped = 0;
    }
}

bool ndt_grid_is_nice(const NdtGrid &g, const double centre[3])
{
    const double res = g.res;
    if ((double)(float)res != res || !(res > 0)) return false;
    // res = odd * 2^e: (m + k) * res is an fp32 number when |m + k| * odd < 2^24
    int e = 0;
    double mant = frexp(res, &e);                 // res = mant * 2^e, mant in [0.5, 1)
    double odd = mant * 16777216.0;               // a 24-bit integer (res is an fp32 number)
    while (odd != 0.0 && fmod(odd, 2.0) == 0.0) odd /= 2.0;
    for (int a = 0; a < 3; a++) {
        const double c0 = centre[a] - (g.size[a] / 2.0) * res;
        const double q = c0 / res;
        if (q != rint(q) || q * res != c0) return false;          // the lattice of cell origins passes through c0 = q res
        if ((fabs(q) + (double)g.size[a] + 1.0) * odd >= 16777216.0) return false;
    }
    return true;
}

hipError_t ndt_launch_build(const NdtSetView &set, size_t first, size_t count, const void *xyz_dev, size_t n_points,
                            size_t stride_bytes, size_t map_stride_bytes, double range_limit,
                            const double *range_origins_dev, int n_min, double eval_factor, int nice, hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    // scales that keep every 64-bit accumulator below 2^62.  Even grid sizes: |u| <= 1/2 (+ rounding), so
    // N/2 * 2^s1 and N/4 * 2^s2 must stay below 2^62.  An odd size lets the reference's double->int truncation put
    // offsets of up to 1.5 cells into index 0: bound |u| < 2 there.
    int lg = 1;
    while ((1ull << lg) < (unsigned long long)(n_points ? n_points : 1)) lg++;
    const bool odd = (set.grid.size[0] | set.grid.size[1] | set.grid.size[2]) & 1;
    int s1_shift = (odd ? 60 : 62) - lg;
    int s2_shift = (odd ? 58 : 62) - lg;
    if (s1_shift > 45) s1_shift = 45;
    if (s2_shift > 45) s2_shift = 45;
    const int dbg = 0;   // reserved kernel argument
    const bool aligned4 = (((uintptr_t)xyz_dev | map_stride_bytes) & 3u) == 0;
    const int sdw = (stride_bytes == 12 && aligned4) ? 3 : (stride_bytes == 16 && aligned4) ? 4 : 0;
    // Few maps: spread each scan over several workgroups (accumulate) and finalise in a second launch.
    const unsigned n_tiles = (unsigned)((n_points + NDT_TILE - 1) / NDT_TILE);
    unsigned parts = 1;
    if (count < 256 && n_tiles > 8) {
        const char *pe = getenv("NDTGPU_BUILD_WGS");             // (experiments: workgroups of the accumulate launch)
        // as many workgroups as are resident at once: three per CU (149 - 161 registers: three waves per SIMD; the LDS of the
        // flat-grid variants would hold four).  64 3D sweeps on 512 / 768 / 1024 / 1536 / 2048 workgroups: 0.68 / 0.64 /
        // 0.78 / 0.68 / 0.70 ms per build; 32 / 64 / 128 planar scans on 768 against 1024: 0.105 / 0.132 / 0.181 against
        // 0.117 / 0.149 / 0.190 ms
        const unsigned wgs = pe && atoi(pe) > 0 ? (unsigned)atoi(pe) : 768u;
        parts = (unsigned)(wgs / count);
        if (parts > n_tiles / 4) parts = n_tiles / 4;
        if (parts < 1) parts = 1;
    }
    // batches of planar scans on a grid whose cell centres are fp32 numbers: the wave-uniform kernel of
    // csrc/ndt_build_flat.hip (NDTGPU_FLAT=0: never, 2: also for the few-maps case, one workgroup per map)
    {
        const char *fe = getenv("NDTGPU_FLAT");
        const int flat_mode = fe ? atoi(fe) : 1;
        if (flat_mode && (parts == 1 || flat_mode == 2) && ndt_build_flat_ok(set.grid, nice, sdw))
            return ndt_launch_build_flat(set, first, count, xyz_dev, n_points, sdw, map_stride_bytes, range_limit,
                                         range_origins_dev, n_min, eval_factor, s1_shift, s2_shift, stream);
    }
    // thick grids hold 3D sweeps, whose consecutive points change cell every few points: replaced runs go to the wide
    // flush list (SCAT).  Flat grids hold planar scans, whose points stay in a cell for hundreds of points.
    const bool scat = set.grid.size[2] > 4;
#define NDT_LAUNCH_BUILD_V(SDW, MODE, NICE_, SCAT_, GRID)                                                            \
    hipLaunchKernelGGL((ndt_build_kernel<SDW, MODE, NICE_, SCAT_>), GRID, dim3(NDT_BUILD_THREADS), 0, stream, set,   \
                       (unsigned)first, (const char *)xyz_dev, (unsigned)n_points, (unsigned)stride_bytes,           \
                       map_stride_bytes, range_limit, range_origins_dev, n_min, eval_factor, s1_shift, s2_shift, dbg, \
                       __builtin_inff())
#define NDT_LAUNCH_BUILD(SDW, MODE, GRID)                                                                            \
    do {                                                                                                             \
        if (nice && scat) NDT_LAUNCH_BUILD_V(SDW, MODE, true, true, GRID);                                           \
        else if (nice) NDT_LAUNCH_BUILD_V(SDW, MODE, true, false, GRID);                                             \
        else if (scat) NDT_LAUNCH_BUILD_V(SDW, MODE, false, true, GRID);                                             \
        else NDT_LAUNCH_BUILD_V(SDW, MODE, false, false, GRID);                                                      \
    } while (0)
#define NDT_LAUNCH_BUILD_SD(MODE, GRID)                                                                              \
    do {                                                                                                             \
        if (sdw == 3) NDT_LAUNCH_BUILD(3, MODE, GRID);                                                               \
        else if (sdw == 4) NDT_LAUNCH_BUILD(4, MODE, GRID);                                                          \
        else NDT_LAUNCH_BUILD(0, MODE, GRID);                                                                        \
    } while (0)
    if (parts == 1) {
        NDT_LAUNCH_BUILD_SD(0, dim3((unsigned)count));
    } else {
        // reset {overflow, n_dropped} of the maps, accumulate on `parts` workgroups per map, finalise
        hipError_t e = hipMemset2DAsync(&set.counters[first].overflow, sizeof(NdtMapCounters), 0, 2 * sizeof(uint32_t),
                                        count, stream);
        if (e != hipSuccess) return e;
        {
            const char *xe = getenv("NDTGPU_BUILD_XCD");           // experiment: bit 1 deals the maps to XCDs (map m -> XCD m mod 8): 1.31 -> 1.28 ms per 64 sweeps
            const int xm = xe ? atoi(xe) : 0;
            const int dbg = (count % 8 == 0) ? xm : 0;
            NDT_LAUNCH_BUILD_SD(1, dim3(parts, (unsigned)count));
#ifdef NDT_BUILD_PROF
            if (xm & 16) return hipGetLastError();                  // (section clocks of phase A stay in the counters)
#endif
        }
        // phases 0 and B (moments -> Gaussians, one per thread) on up to 32 workgroups per map
        // big grids (a 400 x 400 x 40 grid has 200 k bitmap words): the ranking is a third launch on the same number of
        // workgroups per map instead of the last workgroup of the second one walking the whole bitmap alone
        const unsigned bm_words = (unsigned)((set.grid.slots + 31) / 32);
        // Workgroups of the finalise launches (1024 threads, 118 registers: ONE is resident per CU).  Measured, end of round
        // 4: big grids, 64 sweeps on 128 / 256 / 512 / 1024 workgroups per launch: 0.72 / 0.60 / 0.64 / 0.69 ms per build
        // (more segments lengthen the look-back of the ranking launch: 2048 of them 0.77 ms); small grids, where the last
        // workgroup of a map ranks it alone, 8 / 32 / 64 / 128 planar scans on 64 against 512: 0.059 / 0.073 / 0.092 / 0.138 against
        // 0.066 / 0.108 / 0.137 / 0.180 ms.
        const char *fe = getenv("NDTGPU_FIN_WGS");               // (experiments)
        unsigned fin_parts = (unsigned)((fe && atoi(fe) > 0 ? (unsigned)atoi(fe) : (bm_words >= 16384u ? 256u : 64u)) / count);
        if (fin_parts > 32u) fin_parts = 32u;
        if (fin_parts < 1u) fin_parts = 1u;
        const int split_rank = (fin_parts > 1u && bm_words >= 16384u) ? 1 : 0;
        hipLaunchKernelGGL((ndt_build_kernel<0, 2, false, false>), dim3(fin_parts, (unsigned)count), dim3(NDT_FIN_THREADS), 0, stream,
                           set, (unsigned)first, (const char *)xyz_dev, (unsigned)n_points, (unsigned)stride_bytes,
                           map_stride_bytes, range_limit, range_origins_dev, n_min, eval_factor, s1_shift, s2_shift, split_rank,
                           __builtin_inff());
        // (the ranking and the placement launch may take other workgroup counts than the Gaussians: experiments)
        auto parts_of = [&](const char *name, unsigned dflt, unsigned cap) {
            const char *e = getenv(name);
            unsigned p = (e && atoi(e) > 0 ? (unsigned)atoi(e) : dflt) / (unsigned)count;
            return p > cap ? cap : (p < 1u ? 1u : p);
        };
        const unsigned rank_parts = split_rank ? parts_of("NDTGPU_RANK_WGS", fin_parts * (unsigned)count, (unsigned)NDT_RANK_SEGS) : 1u;
        const unsigned place_parts = split_rank ? parts_of("NDTGPU_PLACE_WGS", fin_parts * (unsigned)count, 64u) : 1u;
        if (split_rank)
            hipLaunchKernelGGL((ndt_build_kernel<0, 3, false, false>), dim3(rank_parts, (unsigned)count), dim3(NDT_FIN_THREADS), 0,
                               stream, set, (unsigned)first, (const char *)xyz_dev, (unsigned)n_points, (unsigned)stride_bytes,
                               map_stride_bytes, range_limit, range_origins_dev, n_min, eval_factor, s1_shift, s2_shift, 0,
                               __builtin_inff());
        if (split_rank)
            hipLaunchKernelGGL(ndt_place_cells_kernel, dim3(place_parts, (unsigned)count), dim3(NDT_FIN_THREADS), 0, stream, set,
                               (unsigned)first);
    }
#undef NDT_LAUNCH_BUILD_SD
#undef NDT_LAUNCH_BUILD
#undef NDT_LAUNCH_BUILD_V
    return hipGetLastError();
}

// Phase A only (MODE 1) for the incremental update of csrc/ndt_fuse.hip: the points of one cloud per map are added
// to the moment accumulators of their cells; nothing is finalised.  z_max: NDTMap::addPointCloud's maxz.
hipError_t ndt_launch_accumulate(const NdtSetView &set, size_t first, size_t count, const void *xyz_dev, size_t n_points,
                                 size_t stride_bytes, size_t map_stride_bytes, double range_limit,
                                 const double *range_origins_dev, double z_max, int nice, int *s1_shift_out, int *s2_shift_out,
                                 hipStream_t stream)
{
    int lg = 1;
    while ((1ull << lg) < (unsigned long long)(n_points ? n_points : 1)) lg++;
    const bool odd = (set.grid.size[0] | set.grid.size[1] | set.grid.size[2]) & 1;
    int s1_shift = (odd ? 60 : 62) - lg, s2_shift = (odd ? 58 : 62) - lg;
    if (s1_shift > 45) s1_shift = 45;
    if (s2_shift > 45) s2_shift = 45;
    *s1_shift_out = s1_shift;
    *s2_shift_out = s2_shift;
    if (count == 0 || n_points == 0) return hipSuccess;
    // the largest float that does not exceed z_max: `z <= zf` in float == `(double) z <= z_max`
    float zf = (float)z_max;
    if ((double)zf > z_max) zf = nextafterf(zf, -__builtin_inff());
    const bool aligned4 = (((uintptr_t)xyz_dev | map_stride_bytes) & 3u) == 0;
    const int sdw = (stride_bytes == 12 && aligned4) ? 3 : (stride_bytes == 16 && aligned4) ? 4 : 0;
    const unsigned n_tiles = (unsigned)((n_points + NDT_TILE - 1) / NDT_TILE);
    unsigned parts = (unsigned)(768u / count);   // resident workgroups (see ndt_launch_build)
    if (parts > n_tiles / 4) parts = n_tiles / 4;
    if (parts < 1) parts = 1;
    hipError_t e = hipMemset2DAsync(&set.counters[first].overflow, sizeof(NdtMapCounters), 0, 2 * sizeof(uint32_t), count, stream);
    if (e != hipSuccess) return e;
    const bool scat = set.grid.size[2] > 4;
#define NDT_LAUNCH_ACC_V(SDW, NICE_, SCAT_)                                                                          \
    hipLaunchKernelGGL((ndt_build_kernel<SDW, 1, NICE_, SCAT_>), dim3(parts, (unsigned)count), dim3(NDT_BUILD_THREADS), 0, \
                       stream, set, (unsigned)first, (const char *)xyz_dev, (unsigned)n_points, (unsigned)stride_bytes, \
                       map_stride_bytes, range_limit, range_origins_dev, 0, 0.0, s1_shift, s2_shift, 0, zf)
#define NDT_LAUNCH_ACC(SDW)                                                                                          \
    do {                                                                                                             \
        if (nice && scat) NDT_LAUNCH_ACC_V(SDW, true, true);                                                         \
        else if (nice) NDT_LAUNCH_ACC_V(SDW, true, false);                                                           \
        else if (scat) NDT_LAUNCH_ACC_V(SDW, false, true);                                                           \
        else NDT_LAUNCH_ACC_V(SDW, false, false);                                                                    \
    } while (0)
    if (sdw == 3) NDT_LAUNCH_ACC(3);
    else if (sdw == 4) NDT_LAUNCH_ACC(4);
    else NDT_LAUNCH_ACC(0);
#undef NDT_LAUNCH_ACC
#undef NDT_LAUNCH_ACC_V
    return hipGetLastError();
}

hipError_t ndt_launch_install_cells(const NdtSetView &set, size_t map, const NdtCell *cells_dev, size_t n_cells,
                                    hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(set.rankmap + map * ndt_rm_stride(set.grid), 0, ndt_rm_stride(set.grid) * sizeof(uint2), stream);
    if (e != hipSuccess) return e;
    unsigned blocks = (unsigned)((n_cells + 255) / 256);
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(ndt_install_cells_kernel, dim3(blocks), dim3(256), 0, stream, set, (unsigned)map, cells_dev,
                       (unsigned)n_cells);
    return hipGetLastError();
}
