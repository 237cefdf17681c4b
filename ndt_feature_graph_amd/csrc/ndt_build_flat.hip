// ndt_build_flat.hip -- batch NDT grid build for FLAT grids (planar scans: 2D lidar sweeps) on CDNA4 (gfx950), round 4.
//
// Replaces, like ndt_build.hip (the general kernel, which keeps every other case: thick grids, odd sizes, grids whose
// cell centres are not fp32 numbers, other strides, the few-maps split, the accumulate-only launch of the fuser):
//   LazyGrid::getIndexForPoint / addPoint, NDTMap::loadPointCloud(cloud, range)
//       ndt_feature/src/ndt_feature_src/ndt_feature_fuser_hmt.cpp:195-226
//   NDTMap::computeNDTCells(CELL_UPDATE_MODE_SAMPLE_VARIANCE) -> NDTCell::computeGaussian + rescaleCovariance
//       ...fuser_hmt.cpp:227, ndt_odom_debug.cpp:179
//
// Design.  One 256-thread workgroup per map, four per CU (128 VGPRs, 32 KB of LDS); a wave owns a contiguous quarter of the
// scan and walks it in ROUNDS of 64 CONSECUTIVE points, lane l <-> point 64 r + l (one fully coalesced 12 / 16-byte load
// per lane, six rounds issued together, no staging through LDS).
//   * A planar sweep stays in a cell for hundreds of consecutive points, so the 64 points of a round lie in one cell, or in
//     two when a wall hugs a cell face or the sweep crosses into the next cell.  The wave therefore tracks TWO cells
//     ("runs" A and B) in SCALAR registers -- slot, centre, point count -- and every lane keeps its share of their nine
//     moments (sum d, sum d d^T of d = p - cell centre) in fp64 registers.
//   * Membership of a point in run A is |p - centreA| < res (1/2 - guard) on the three axes: on a grid whose cell centres
//     are fp32 numbers the difference is exact, so this decides exactly what the reference's floor((p - c)/res + 0.5)
//     decides for every point outside a 4e-6-cell band at the faces -- 6 instructions instead of a 40-instruction index.
//     A round whose 64 points all pass for A or B costs 12 instructions of tests + 12 per run that received points.
//     Points that are certainly dropped (NaN, far out of range / grid: beams without a return) count as passed.
//   * A round with a point that fails everything (a new cell, a point in the guard band) is binned with the
//     reference-exact index arithmetic of csrc/ndt_binning.h; a new cell becomes run B, what was B becomes A, and what
//     was A -- the run opened first -- leaves: its moments are summed over the wave (a value-halving butterfly: 57
//     instructions for the nine sums), converted to 64-bit fixed point and appended to the wave's flush list.  One
//     record per cell VISIT of a wave.  The rounds of a batch are processed in scan order by unrolled code with this
//     path inline; the kernel's code (40 KB) has to stay within the 64 KB instruction cache two CUs share (measured:
//     +10 % time beyond it).  The kernel issues vector instructions 80 % of the time its waves are resident: what
//     counts is the number of instructions per round and per visit (DESIGN.md 4.1a has the census).
//   * The list is drained 16 records at a time: slot -> accumulator id through a per-workgroup LDS hash (the map belongs
//     to this workgroup: no global work table), then one 64-bit integer atomic per (record, moment) into the map's
//     accumulators in L2 -- exact, hence order-independent and bit-reproducible, like the general kernel.
//   * Finalise: the hash is the list of touched cells (compacted in place); moments -> Gaussian per cell (shared with the
//     general kernel), Gaussian cells into an LDS bitmap, ranks by popcount prefix, the rank map written DENSELY (every
//     word), cell records written in slot order.  No dense slot -> rank table, no global work table or bitmap: HBM traffic
//     is the points, the accumulators (dense by id) and the outputs.
#include "ndt_math.h"
#include "ndt_binning.h"
#include "ndt_wave.h"
#include <algorithm>

#ifndef NDT_FLAT_THREADS
#define NDT_FLAT_THREADS 256
#endif
#define NDT_FLAT_WAVES (NDT_FLAT_THREADS / 64)
#ifndef NDT_FLAT_U
#define NDT_FLAT_U 6             // rounds of 64 points a wave has in flight while it works on the previous ones
#endif
#ifndef NDT_FLAT_WPE
#define NDT_FLAT_WPE 4            // waves per SIMD the register budget is cut for (four 256-thread workgroups per CU)
#endif
#ifndef NDT_FLAT_SINGLE
#define NDT_FLAT_SINGLE 1         // 1: a batch is loaded, awaited, worked on (the other waves of the SIMD cover the wait); 0: double buffer
#endif
#ifndef NDT_FLAT_LDSRED
#define NDT_FLAT_LDSRED 0         // 1: a run's moments are summed over the wave through LDS (below: 14 additions instead of 57
                                  // instructions, but three dependent LDS round trips -- measured 1.07 against 1.03 ms per 2048
                                  // scans, the exact path is latency bound); 0: in registers (butterfly)
#endif
#define NDT_FLAT_RED_DOUBLES (9 * 64)   // LDS scratch of one wave's sum: nine moments x 64 lanes
#ifndef NDT_FLAT_GONE
#define NDT_FLAT_GONE 1           // points that are certainly dropped (NaN, far out of range / grid) do not send a round to the
                                  // exact path.  Per 2048 scans, clean / a fifth of the beams NaN scattered: with it 1.00 / 1.13 ms,
                                  // without 0.99 / 1.27 ms (U = 6).  The code must stay within the 64 KB instruction cache that two
                                  // CUs share: with U = 8 the same switch costs 1.08 / 1.22 against 0.97 / 1.28
#endif
#define NDT_FLAT_LIST 16         // records in a wave's flush list
#define NDT_FLAT_IDBITS 13       // hash entry = (slot + 1) << 13 | accumulator id

namespace {

NDT_D void flat_acc(double (&sd)[3], double (&se)[6], float dx, float dy, float dz)
{
    const double x = (double)dx, y = (double)dy, z = (double)dz;
    sd[0] += x; sd[1] += y; sd[2] += z;
    se[0] = fma(x, x, se[0]); se[1] = fma(x, y, se[1]); se[2] = fma(x, z, se[2]);
    se[3] = fma(y, y, se[3]); se[4] = fma(y, z, se[4]); se[5] = fma(z, z, se[5]);
}

NDT_D float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// slot -> accumulator id through the workgroup's LDS hash (open addressing, linear probing), allocating on first touch.
// Lock-free: the id is drawn before the entry is claimed; a lane that loses the race for a slot uses the winner's id and
// its own is wasted (left with n == 0, skipped by the finaliser) -- as in the general kernel's global work table.
NDT_D int flat_id_of(unsigned *hash, unsigned hash_mask, unsigned hash_shift, unsigned *nalloc, unsigned cap,
                     unsigned *ovf, int slot)
{
    const unsigned key = (unsigned)slot + 1u;
    unsigned h = (key * 0x9E3779B1u) >> hash_shift;
    int drawn = -1;
    for (unsigned probes = 0; probes <= hash_mask; probes++) {
        unsigned e = __hip_atomic_load(&hash[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (e == 0u) {
            if (drawn < 0) {
                drawn = (int)__hip_atomic_fetch_add(nalloc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((unsigned)drawn >= cap) { *ovf = 1u; return -1; }
            }
            unsigned expected = 0u;
            const unsigned want = (key << NDT_FLAT_IDBITS) | (unsigned)drawn;
            if (__hip_atomic_compare_exchange_strong(&hash[h], &expected, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP))
                return drawn;
            e = expected;
        }
        if ((e >> NDT_FLAT_IDBITS) == key) return (int)(e & ((1u << NDT_FLAT_IDBITS) - 1u));
        h = (h + 1u) & hash_mask;
    }
    *ovf = 1u;
    return -1;
}

}  // namespace

// SD: dwords per point record (3 = packed xyz, 4 = pcl::PointXYZ)
template <int SD>
__global__ __launch_bounds__(NDT_FLAT_THREADS) __attribute__((amdgpu_waves_per_eu(NDT_FLAT_WPE, NDT_FLAT_WPE))) void ndt_build_flat_kernel(
    NdtSetView set, unsigned first, const char *__restrict__ xyz, unsigned n_points, size_t map_stride_bytes,
    double range_limit, const double *__restrict__ range_origins, int n_min, double eval_factor, int s1_shift, int s2_shift,
    unsigned hash_log2)
{
    extern __shared__ __attribute__((aligned(16))) unsigned s_dyn[];
    __shared__ long long s_lval[NDT_FLAT_WAVES * NDT_FLAT_LIST * 10];
    __shared__ int s_lslot[NDT_FLAT_WAVES * NDT_FLAT_LIST];
    __shared__ int s_lid[NDT_FLAT_WAVES * NDT_FLAT_LIST];
    __shared__ unsigned s_wave_cnt[NDT_FLAT_WAVES];
    __shared__ unsigned s_nalloc, s_binned, s_ovf;
#ifdef NDT_FLAT_STATS
    __shared__ unsigned s_stat[4];   // rounds that failed the first fast test, exact-path rounds, flushes, drains
    if (threadIdx.x < 4) s_stat[threadIdx.x] = 0u;
#define NDT_FLAT_STAT(k) do { if (lane == 0) atomicAdd(&s_stat[k], 1u); } while (0)
#else
#define NDT_FLAT_STAT(k) do { } while (0)
#endif

    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63u, wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const unsigned map_local = blockIdx.x, map = first + map_local;
    const NdtGrid g = set.grid;
    const uint32_t cap = g.max_cells;
    const unsigned bm_words = (unsigned)((g.slots + 31) >> 5);
    const unsigned hash_entries = 1u << hash_log2, hash_mask = hash_entries - 1u, hash_shift = 32u - hash_log2;
    // region 0: the waves' reduction scratch during phase A (NDT_FLAT_LDSRED), the Gaussian-cell bitmap from phase B on
    const unsigned region0_words = NDT_FLAT_LDSRED ? max(bm_words, (unsigned)(NDT_FLAT_WAVES * NDT_FLAT_RED_DOUBLES * 2)) : bm_words;
    unsigned *s_bits = s_dyn;                      // [bm_words]   Gaussian-cell bit per slot (phase B on)
    [[maybe_unused]] double *s_red = reinterpret_cast<double *>(s_dyn);   // [waves][9][64] (phase A)
    unsigned *s_hash = s_dyn + region0_words;      // [hash_entries] (slot + 1) << 13 | id; compacted per wave after phase A

    uint2 *rankmap = set.rankmap + (size_t)map * ndt_rm_stride(g);
    NdtCell *cells = set.cells + (size_t)map * cap;
    NdtAcc *acc = set.acc + (size_t)map * cap;
    NdtMapCounters *ctr = set.counters + map;
    const double cx = set.centres[map * 3 + 0], cy = set.centres[map * 3 + 1], cz = set.centres[map * 3 + 2];
    const double res = g.res, inv_res = 1.0 / g.res;
    const double hx = g.half[0], hy = g.half[1], hz = g.half[2];      // (kernel arguments: scalar registers)
    double ox = 0, oy = 0, oz = 0;
    if (range_origins) { ox = range_origins[map_local * 3]; oy = range_origins[map_local * 3 + 1]; oz = range_origins[map_local * 3 + 2]; }
    const char *pts = xyz + (size_t)map_local * map_stride_bytes;

    for (unsigned i = tid; i < hash_entries; i += NDT_FLAT_THREADS) s_hash[i] = 0u;
    if (tid == 0) { s_nalloc = 0u; s_binned = 0u; s_ovf = 0u; }
    __syncthreads();

    // ---------------- phase A: the scan, once ---------------------------------------------------------------------------
    const long long t0 = __builtin_readcyclecounter();
#ifdef NDT_FLAT_TIMES
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
#endif
    {
        NdtBinner bn;
        bn.init(g, cx, cy, cz, ox, oy, oz, range_limit, __builtin_inff());
        bn.scalarize();
        // cell centres are fp32 numbers (the launcher checked ndt_grid_is_nice for every map of the launch)
        // (uniform_f: wave-uniform floats come out of the vector ALU; a readfirstlane moves them to scalar registers)
        const float res32 = uniform_f((float)res);
        const float c0x32 = uniform_f((float)(cx - hx * res)), c0y32 = uniform_f((float)(cy - hy * res)),
                    c0z32 = uniform_f((float)(cz - hz * res));
        const float ox32 = uniform_f((float)ox), oy32 = uniform_f((float)oy), oz32 = uniform_f((float)oz);
        const float half32 = uniform_f(0.5f * res32);
        const float lim_in = uniform_f(res32 * (0.5f - 4e-6f));   // |p - centre| below this on every axis: in the cell, exactly
        const float r2safe = uniform_f(range_limit > 0 ? (float)(range_limit * range_limit) * (1.0f - 2e-3f) : __builtin_inff());
        // which moment of a flushed run this lane hands to the list, and its scale to fixed point
#if NDT_FLAT_LDSRED
        // (lanes 0..8 end up with the totals of moments 0..8)
        const int my_moment = lane < 9u ? (int)lane : -1;
        double *red = s_red + wave * NDT_FLAT_RED_DOUBLES;
        const unsigned red_v = lane / 7u, red_s = lane - red_v * 7u;          // stage 2: lane (v, s) sums 9 lanes' values of moment v
#else
        const int my_moment = ndt_moment_of_lane(lane);
#endif
        const double my_scale = my_moment < 3 ? ldexp(inv_res, s1_shift) : ldexp(inv_res * inv_res, s2_shift);

        long long *lval = s_lval + wave * (NDT_FLAT_LIST * 10);
        int *lslot = s_lslot + wave * NDT_FLAT_LIST, *lid = s_lid + wave * NDT_FLAT_LIST;
        unsigned nfl = 0;                                      // records in the list (wave-uniform)

        // the two runs.  Scalars (wave-uniform): slot (negative = empty), cell centre, membership limit, points.
        int a_slot = -2, b_slot = -3;
        float a_cx = 0, a_cy = 0, a_cz = 0, b_cx = 0, b_cy = 0, b_cz = 0;
        float a_lim = -1.0f, b_lim = -1.0f;                    // -1: no point passes the fast test (empty run, or a cell
                                                               // that is not entirely inside the range sphere)
        unsigned a_n = 0, b_n = 0;
        bool seen_gone = false;                                // the exact path has dropped a point of this wave's share
        double a_sd[3] = {0, 0, 0}, a_se[6] = {0, 0, 0, 0, 0, 0}, b_sd[3] = {0, 0, 0}, b_se[6] = {0, 0, 0, 0, 0, 0};

        auto drain = [&]() {
            NDT_FLAT_STAT(3);
            ndt_wave_sync();                                   // the records were written by other lanes
            if (lane < nfl) {
                const int id = flat_id_of(s_hash, hash_mask, hash_shift, &s_nalloc, cap, &s_ovf, lslot[lane]);
                lid[lane] = id;
            }
            ndt_wave_sync();
            const unsigned items = nfl * 10u;
            for (unsigned it = lane; it < items; it += 64u) {
                const unsigned e = it / 10u, k = it - e * 10u;
                const int id = lid[e];
                if (id >= 0) atomicAdd(reinterpret_cast<unsigned long long *>(acc + id) + k, (unsigned long long)lval[it]);
            }
            ndt_wave_sync();                                   // the list may be overwritten now
            nfl = 0;
        };
        // the moments of a run, summed over the wave, become one record of the flush list; the run's registers go to zero
        auto flush = [&](double (&sd)[3], double (&se)[6], int slot, unsigned n) {
            NDT_FLAT_STAT(2);
#if defined(NDT_FLAT_ABL) && NDT_FLAT_ABL == 1
            const double t = sd[0] + se[5];      // (ablation: no sum over the wave -- wrong results, timing only)
#elif NDT_FLAT_LDSRED
            // Sum over the 64 lanes through LDS, in a fixed order: every lane stores its nine values (moment-major: no
            // bank conflict), 63 lanes add nine neighbours each (the last segment ten), nine lanes add the seven segment
            // sums.  14 additions in the vector ALU instead of the 57 instructions of the register butterfly
            // (csrc/ndt_wave.h), and three LDS round trips instead of its six dependent levels.
#pragma unroll
            for (int k = 0; k < 3; k++) red[k * 64 + lane] = sd[k];
#pragma unroll
            for (int k = 0; k < 6; k++) red[(3 + k) * 64 + lane] = se[k];
            ndt_wave_sync();
            double part = 0.0;
            if (lane < 63u) {
                const double *seg = red + red_v * 64u + red_s * 9u;
                part = seg[0];
#pragma unroll
                for (int i = 1; i < 9; i++) part += seg[i];
                if (red_s == 6u) part += seg[9];                       // (lane 63's value: the last segment has ten)
            }
            ndt_wave_sync();                                           // everybody has read: the area is reused
            if (lane < 63u) red[lane] = part;
            ndt_wave_sync();
            double t = 0.0;
            if (lane < 9u) {
                const double *q = red + lane * 7u;
                t = q[0];
#pragma unroll
                for (int i = 1; i < 7; i++) t += q[i];
            }
            ndt_wave_sync();
#else
            const double t = wave_sum_moments(sd, se, lane);
#endif
            if (nfl == NDT_FLAT_LIST) drain();
            if (my_moment >= 0) lval[nfl * 10u + 1u + (unsigned)my_moment] = ndt_fixed_from_double(t * my_scale);
            if (lane == 9) { lval[nfl * 10u] = (long long)n; lslot[nfl] = slot; }
            nfl++;
#pragma unroll
            for (int k = 0; k < 3; k++) sd[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) se[k] = 0.0;
        };
        // membership limit of a cell with centre c: the fast test only serves cells that lie entirely inside the range
        // sphere (with the margin of the binner's fp64 band), so that it never has to look at the range
        auto cell_lim = [&](float ccx, float ccy, float ccz) -> float {
            const float fx = fabsf(ccx - ox32) + half32, fy = fabsf(ccy - oy32) + half32, fz = fabsf(ccz - oz32) + half32;
            const float far2 = fx * fx + fy * fy + fz * fz;
            return far2 < r2safe ? lim_in : -1.0f;
        };

        // One round through the fast tests.  Returns true when some point passes neither (nothing was added then).
        // Run B is the cell the sweep entered last: a round that lies entirely in it -- the usual case -- is done after
        // three compares.  (Single exit, accumulators updated in place under the lane masks at ONE site per run: the
        // register allocator then keeps one copy of the 36 accumulator registers; a version with early returns was
        // compiled with three.)
        auto fast_round = [&](float px, float py, float pz, unsigned long long m_gone) -> bool {
            const float bx = px - b_cx, by = py - b_cy, bz = pz - b_cz;
            // (three compares, not a compare of the maximum: v_max3_f32 drops a NaN operand, and a point with one NaN
            //  coordinate must fail)
            const bool in_b = (fabsf(bx) < b_lim) & (fabsf(by) < b_lim) & (fabsf(bz) < b_lim);
            const unsigned long long m_b = ndt_ballot(in_b);
            const float ax = px - a_cx, ay = py - a_cy, az = pz - a_cz;
            bool in_a = false;
            unsigned long long m_a = 0ull;
            bool all = m_b == ~0ull;
            if (!all) {
                in_a = (fabsf(ax) < a_lim) & (fabsf(ay) < a_lim) & (fabsf(az) < a_lim);
                m_a = ndt_ballot(in_a);
                all = (m_a | m_b | m_gone) == ~0ull;               // m_gone: lanes whose point is certainly dropped
            }
            if (all) {
                if (m_a) {
                    if (in_a) flat_acc(a_sd, a_se, ax, ay, az);
                }
                if (m_b) {
                    if (in_b) flat_acc(b_sd, b_se, bx, by, bz);
                }
                a_n += (unsigned)__popcll(m_a);
                b_n += (unsigned)__popcll(m_b);
            }
            return !all;
        };
        // Points that are CERTAINLY dropped, without the index arithmetic: NaN coordinates, points beyond the range sphere by
        // more than the binner's band, points more than a hundredth of a cell outside the grid's box.  A real scan is
        // full of them (beams without a return), and a round must not take the exact path because of them.
        const float r2gone = range_limit > 0 ? uniform_f((float)(range_limit * range_limit) * (1.0f + 2e-3f)) : __builtin_inff();
        const float glo_x = uniform_f(c0x32 - 0.51f * res32), ghi_x = uniform_f(c0x32 + ((float)g.size[0] - 0.49f) * res32);
        const float glo_y = uniform_f(c0y32 - 0.51f * res32), ghi_y = uniform_f(c0y32 + ((float)g.size[1] - 0.49f) * res32);
        const float glo_z = uniform_f(c0z32 - 0.51f * res32), ghi_z = uniform_f(c0z32 + ((float)g.size[2] - 0.49f) * res32);
        auto gone_mask = [&](float px, float py, float pz) -> unsigned long long {
            const float dx = px - ox32, dy = py - oy32, dz = pz - oz32;
            const bool in_box = (px > glo_x) & (px < ghi_x) & (py > glo_y) & (py < ghi_y) & (pz > glo_z) & (pz < ghi_z);   // false for NaN
            const bool far = dx * dx + dy * dy + dz * dz > r2gone;
            return ndt_ballot(!in_box | far);
        };
        // The same round with the reference's index arithmetic.  A new cell becomes run B; the cell that was B becomes A
        // and what was A -- the run that was opened first -- leaves through the flush list: first in, first out.  (Least
        // recently USED was the first policy: 7 vector instructions per round to keep track of it, and a victim that is
        // A or B -- two copies of the flush, or 72 register moves per new cell to make it always A.  On a sweep the
        // cell entered first is the one left behind: 410 flushes per scan of the bench's rooms with either policy.)
        auto slow_round = [&](float px, float py, float pz) {
            float gx, gy, gz;
            int slot;
            const bool near = bn.fast(px, py, pz, gx, gy, gz, slot);
            if (ndt_ballot(near)) {
                if (near) bn.exact(px, py, pz, gx, gy, gz, slot);
            }
            seen_gone = seen_gone || ndt_ballot(slot < 0) != 0ull;
            {
                const bool in_a = slot == a_slot, in_b = slot == b_slot;       // (empty runs: -2 / -3, dropped points: -1)
                const unsigned long long m_a = ndt_ballot(in_a), m_b = ndt_ballot(in_b);
                if (m_a) {
                    if (in_a) flat_acc(a_sd, a_se, px - a_cx, py - a_cy, pz - a_cz);
                    a_n += (unsigned)__popcll(m_a);
                }
                if (m_b) {
                    if (in_b) flat_acc(b_sd, b_se, px - b_cx, py - b_cy, pz - b_cz);
                    b_n += (unsigned)__popcll(m_b);
                }
                slot = (in_a | in_b) ? -1 : slot;
            }
            // points of other cells: the first one's cell becomes run B, what was B becomes A, what was A is flushed.  (The
            // loop starts with the flush and ends with the new run's first points: the runs' registers change hands
            // once per new cell, at the bottom, on the way to the exit and to the next pass alike.)
            unsigned long long m_s = ndt_ballot(slot >= 0);
            while (m_s) {
                const int lead = __ffsll((long long)m_s) - 1;
                const int c_slot = __builtin_amdgcn_readlane(slot, lead);
                const float c_cx = uniform_f(fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(gx), lead)), res32, c0x32));
                const float c_cy = uniform_f(fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(gy), lead)), res32, c0y32));
                const float c_cz = uniform_f(fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(gz), lead)), res32, c0z32));
                const float c_lim = uniform_f(cell_lim(c_cx, c_cy, c_cz));
                if (a_slot >= 0) flush(a_sd, a_se, a_slot, a_n);
                const bool in_c = slot == c_slot;
                const unsigned long long m_c = ndt_ballot(in_c);
                // the new run's first points: d = p - centre where the lane's point is in the cell, 0 elsewhere
                const float dx = in_c ? px - c_cx : 0.0f, dy = in_c ? py - c_cy : 0.0f, dz = in_c ? pz - c_cz : 0.0f;
                const double x = (double)dx, y = (double)dy, z = (double)dz;
#pragma unroll
                for (int k = 0; k < 3; k++) a_sd[k] = b_sd[k];
#pragma unroll
                for (int k = 0; k < 6; k++) a_se[k] = b_se[k];
                b_sd[0] = x; b_sd[1] = y; b_sd[2] = z;
                b_se[0] = x * x; b_se[1] = x * y; b_se[2] = x * z; b_se[3] = y * y; b_se[4] = y * z; b_se[5] = z * z;
                a_slot = b_slot; a_cx = b_cx; a_cy = b_cy; a_cz = b_cz; a_lim = b_lim; a_n = b_n;
                b_slot = c_slot; b_cx = c_cx; b_cy = c_cy; b_cz = c_cz; b_lim = c_lim; b_n = (unsigned)__popcll(m_c);
                slot = in_c ? -1 : slot;
                m_s = ndt_ballot(slot >= 0);
            }
        };

        const unsigned rounds_total = (n_points + 63u) / 64u;
        const unsigned rounds_per_wave = (rounds_total + NDT_FLAT_WAVES - 1u) / NDT_FLAT_WAVES;
        const unsigned r_begin = min(rounds_total, wave * rounds_per_wave), r_end = min(rounds_total, r_begin + rounds_per_wave);
        struct __attribute__((packed, aligned(4))) P3 { float x, y, z; };
        constexpr int U = NDT_FLAT_U;
        // The points of U rounds: the batch in flight and the batch being worked on.  The rounds of a batch are processed
        // strictly in scan order by UNROLLED code, the exact path inline in every copy: static register names, no replay
        // of a round, no select of the round's registers.
        float nx[U], ny[U], nz[U];
        // Unconditional loads: a wave-uniform base (round clamped to the scan's last one) + a per-lane byte offset, which for
        // the last, possibly partial, round is the offset of the lane clamped to the last point.  (With the loads under
        // lane masks the compiler cannot count how many are in flight and waits for nearly all of them.)
        const unsigned last_round = rounds_total ? rounds_total - 1u : 0u;
        const unsigned tail = n_points - last_round * 64u;                 // points of the last round (1..64)
        const unsigned lane_off = lane * (SD * 4u), lane_off_tail = min(lane, tail - 1u) * (SD * 4u);
        auto load_batch = [&](unsigned r0) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const unsigned r = min(r0 + (unsigned)u, last_round);
                const char *base = pts + (size_t)r * (64u * SD * 4u);
                const P3 p = *reinterpret_cast<const P3 *>(base + (r == last_round ? lane_off_tail : lane_off));
                nx[u] = p.x; ny[u] = p.y; nz[u] = p.z;
            }
        };
#if NDT_FLAT_SINGLE
        // single buffer: a batch is loaded, awaited, worked on; the other waves of the SIMD cover the wait
#pragma unroll 1
        for (unsigned r0 = r_begin; r0 < r_end; r0 += (unsigned)U) {
            load_batch(r0);
            float (&qx)[U] = nx, (&qy)[U] = ny, (&qz)[U] = nz;
#else
        if (r_begin < r_end) load_batch(r_begin);
#pragma unroll 1
        for (unsigned r0 = r_begin; r0 < r_end; r0 += (unsigned)U) {
            float qx[U], qy[U], qz[U];
#pragma unroll
            for (int u = 0; u < U; u++) { qx[u] = nx[u]; qy[u] = ny[u]; qz[u] = nz[u]; }
            load_batch(r0 + (unsigned)U);                      // in flight while this batch is worked on
#endif
            const unsigned nr = min((unsigned)U, r_end - r0);
#pragma unroll
            for (int u = 0; u < U; u++) {
                if ((unsigned)u < nr) {
                    float px = qx[u];
                    if (r0 + (unsigned)u == last_round) px = lane < tail ? px : __builtin_nanf("");   // past the end: NaN points
                    bool todo = fast_round(px, qy[u], qz[u], 0ull);
#if NDT_FLAT_GONE
                    if (todo && seen_gone) {           // (a scan that has not dropped a point so far does not pay for the test)
                        const unsigned long long m_gone = gone_mask(px, qy[u], qz[u]);
                        if (m_gone) todo = fast_round(px, qy[u], qz[u], m_gone);
                    }
#endif
                    if (todo) {
                        NDT_FLAT_STAT(1);
#ifdef NDT_FLAT_STATS
                        const long long ts = __builtin_readcyclecounter();
#endif
                        slow_round(px, qy[u], qz[u]);
#ifdef NDT_FLAT_STATS
                        if (lane == 0) atomicAdd(&s_stat[0], (unsigned)(__builtin_readcyclecounter() - ts) >> 2);   // mean over the 4 waves
#endif
                    }
                }
            }
        }
        if (a_slot >= 0) flush(a_sd, a_se, a_slot, a_n);
        if (b_slot >= 0) flush(b_sd, b_se, b_slot, b_n);
        if (nfl) drain();
#ifdef NDT_FLAT_TIMES
        if (lane == 0) s_wave_cnt[wave] = (unsigned)(__builtin_readcyclecounter() - t0);
#endif
    }
    __syncthreads();
#ifdef NDT_FLAT_TIMES
    unsigned wclk_max = 0, wclk_sum = 0;
    for (int k = 0; k < NDT_FLAT_WAVES; k++) { wclk_max = max(wclk_max, s_wave_cnt[k]); wclk_sum += s_wave_cnt[k]; }
    __syncthreads();
#endif

    // ---------------- phase B: moments -> Gaussian ----------------------------------------------------------------------
    const long long t1 = __builtin_readcyclecounter();
    for (unsigned i = tid; i < bm_words; i += NDT_FLAT_THREADS) s_bits[i] = 0u;      // (region 0 was the waves' scratch until here)
    // the hash IS the list of touched cells: every wave compacts its quarter of the entries in place (a wave reads 64
    // entries before it writes the survivors further down: the write index never passes the read index)
    const unsigned seg_len = hash_entries / NDT_FLAT_WAVES;
    {
        unsigned *seg = s_hash + wave * seg_len;
        unsigned out = 0;
        for (unsigned i = 0; i < seg_len; i += 64u) {
            const unsigned e = seg[i + lane];
            const unsigned long long m = ndt_ballot(e != 0u);
            const unsigned before = (unsigned)__popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64u - lane))));
            ndt_wave_sync();
            if (e != 0u) seg[out + before] = e;
            out += (unsigned)__popcll(m);
        }
        if (lane == 0) s_wave_cnt[wave] = out;
    }
    __syncthreads();
    unsigned seg_cnt[NDT_FLAT_WAVES], n_touched = 0;
#pragma unroll
    for (int k = 0; k < NDT_FLAT_WAVES; k++) { seg_cnt[k] = s_wave_cnt[k]; n_touched += seg_cnt[k]; }
    auto entry_at = [&](unsigned i) -> unsigned {          // i-th touched cell
        unsigned k = 0;
#pragma unroll
        for (int w = 0; w < NDT_FLAT_WAVES - 1; w++)
            if (k == (unsigned)w && i >= seg_cnt[w]) { i -= seg_cnt[w]; k++; }
        return s_hash[k * seg_len + i];
    };
    __syncthreads();                                           // (s_wave_cnt is reused below)
    const double IS1 = ldexp(1.0, -s1_shift), IS2 = ldexp(1.0, -s2_shift);
    NdtCell mine;                                              // the record of this thread's first cell stays in registers
    mine.n = 0;
    unsigned binned = 0;
    for (unsigned i = tid; i < n_touched; i += NDT_FLAT_THREADS) {
        const unsigned e = entry_at(i);
        const unsigned id = e & ((1u << NDT_FLAT_IDBITS) - 1u), slot = (e >> NDT_FLAT_IDBITS) - 1u;
        // the accumulators were updated by atomics, at the L2: read them there (agent scope), past this CU's L1
        NdtAcc a;
        unsigned long long *aw = reinterpret_cast<unsigned long long *>(acc + id);
#pragma unroll
        for (int k = 0; k < 10; k++)
            reinterpret_cast<unsigned long long *>(&a)[k] = __hip_atomic_load(aw + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long n = (unsigned long long)a.n;
        binned += (unsigned)n;
        if (set.occ && n > 0) {
            // NDTCell::computeGaussian on a fresh cell: occ = n log(0.6 / 0.4), clamped to the default limit 255
            const float o = (float)((double)n * NDT_LOGODD_OCC);
            set.occ[(size_t)map * g.slots + slot] = o > 255.0f ? 255.0f : o;
        }
        const int iz = slot % g.size[2], iy = (slot / g.size[2]) % g.size[1], ix = slot / (g.size[2] * g.size[1]);
        const double centre[3] = {cx + (ix - hx) * res, cy + (iy - hy) * res, cz + (iz - hz) * res};
        const NdtCell c = ndt_gaussian_from_moments(a, slot, centre, res, n_min, eval_factor, IS1, IS2);
        if (c.n) atomicOr(&s_bits[slot >> 5], 1u << (slot & 31u));
        if (i < NDT_FLAT_THREADS) {
            mine = c;
#pragma unroll
            for (int k = 0; k < 10; k++) aw[k] = 0ull;         // the accumulator rests at zero between builds
        } else {
            *reinterpret_cast<NdtCell *>(acc + id) = c;        // the record waits in its own accumulator
        }
    }
    if (binned) atomicAdd(&s_binned, binned);
    __syncthreads();

    // ---------------- phase C: ranks in slot order, rank map, cell records ------------------------------------------------
    const long long t2 = __builtin_readcyclecounter();
    const unsigned words_per_wave = ((bm_words + NDT_FLAT_WAVES - 1u) / NDT_FLAT_WAVES + 63u) & ~63u;
    const unsigned wb = min(bm_words, wave * words_per_wave), we = min(bm_words, wb + words_per_wave);
    {
        unsigned cnt = 0;
        for (unsigned w = wb + lane; w < we; w += 64u) cnt += (unsigned)__popc(s_bits[w]);
        const unsigned incl = ndt_wave_incl_scan(cnt);
        if (lane == 63) s_wave_cnt[wave] = incl;
    }
    __syncthreads();
    unsigned running = 0, total_cells = 0;
    for (unsigned k = 0; k < NDT_FLAT_WAVES; k++) {
        const unsigned c2 = s_wave_cnt[k];
        if (k < wave) running += c2;
        total_cells += c2;
    }
    for (unsigned step = wb; step < we; step += 64u) {
        const unsigned w = step + lane;
        const unsigned bits = (w < we) ? s_bits[w] : 0u;
        const unsigned cnt = (unsigned)__popc(bits);
        const unsigned incl = ndt_wave_incl_scan(cnt);
        const unsigned before = running + incl - cnt;
        running += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
        // every word of the map is written: nothing of a previous build survives, nothing has to be forgotten first
        if (w < we) rankmap[w] = make_uint2(bits, before);
    }
    __syncthreads();                                           // (the stores above have left the CU: write-through)
    for (unsigned i = tid; i < n_touched; i += NDT_FLAT_THREADS) {
        NdtCell c;
        if (i < NDT_FLAT_THREADS) {
            c = mine;
        } else {
            const unsigned id = entry_at(i) & ((1u << NDT_FLAT_IDBITS) - 1u);
            c = *reinterpret_cast<const NdtCell *>(acc + id);
            unsigned long long *aw = reinterpret_cast<unsigned long long *>(acc + id);
#pragma unroll
            for (int k = 0; k < 10; k++) aw[k] = 0ull;
        }
        if (c.n) {
            // the rank of the word's first cell was written by another lane: read it at the L2
            const unsigned w = c.slot >> 5, b = c.slot & 31u;
            const unsigned first_rank = __hip_atomic_load(&rankmap[w].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cells[first_rank + (unsigned)__popc(s_bits[w] & ((1u << b) - 1u))] = c;
        }
    }
    const long long t3 = __builtin_readcyclecounter();
    if (tid == 0) {
        ctr->n_cells = total_cells;
        ctr->n_alloc = 0;
        ctr->overflow = s_ovf;
        ctr->n_dropped = n_points - s_binned;
        ctr->cyc[0] = (uint32_t)(t1 - t0);
        ctr->cyc[1] = (uint32_t)(t2 - t1);
        ctr->cyc[2] = (uint32_t)(t3 - t2);
        ctr->cyc[3] = 0u;
#ifdef NDT_FLAT_STATS
        for (int k = 0; k < 4; k++) ctr->cyc[k] = s_stat[k];
#endif
#ifdef NDT_FLAT_TIMES
        // (timeline of the launch: start and end of this workgroup on the 100 MHz clock, its core clocks, where it ran)
        ctr->cyc[0] = (uint32_t)rt0;
        ctr->cyc[1] = (uint32_t)__builtin_amdgcn_s_memrealtime();
        ctr->cyc[2] = wclk_max;            // phase A: the slowest wave's clocks ...
        ctr->cyc[3] = wclk_sum / NDT_FLAT_WAVES;   // ... and the mean of the waves
#endif
        if (set.cell_sel) set.cell_sel[map] = 0u;
    }
}

// whether ndt_launch_build may hand a batch to the flat kernel
bool ndt_build_flat_ok(const NdtGrid &g, int nice, int sdw)
{
    const bool odd = ((g.size[0] | g.size[1] | g.size[2]) & 1) != 0;
    const unsigned bm_words = (unsigned)((g.slots + 31) / 32);
    return nice && !odd && (sdw == 3 || sdw == 4) && g.size[2] <= 4 && g.max_cells <= 4096u &&
           (unsigned)g.slots + 2u < (1u << (32 - NDT_FLAT_IDBITS)) && bm_words <= 4096u;
}

hipError_t ndt_launch_build_flat(const NdtSetView &set, size_t first, size_t count, const void *xyz_dev, size_t n_points,
                                 int sdw, size_t map_stride_bytes, double range_limit, const double *range_origins_dev,
                                 int n_min, double eval_factor, int s1_shift, int s2_shift, hipStream_t stream)
{
    const NdtGrid &g = set.grid;
    const unsigned bm_words = (unsigned)((g.slots + 31) / 32);
    unsigned hash_log2 = 6;                      // (>= 64 entries per wave: the in-place compaction reads whole waves)
    while ((1u << hash_log2) < 64u * NDT_FLAT_WAVES) hash_log2++;
    // (entries >= cells the map may hold: at the usual few hundred to two thousand cells of a 4096-cell map the table is
    //  at most half full; a map that runs into max_cells probes long chains and is flagged as overflowing anyway)
    while ((1u << hash_log2) < g.max_cells) hash_log2++;
    const size_t region0 = NDT_FLAT_LDSRED ? std::max<size_t>(bm_words, (size_t)NDT_FLAT_WAVES * NDT_FLAT_RED_DOUBLES * 2) : bm_words;
    const size_t dyn = (region0 + ((size_t)1 << hash_log2)) * sizeof(unsigned);
    // (static + dynamic LDS of the largest configuration exceed the 64 KB a workgroup gets by default)
    static bool attr_set[2][16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const void *fn = sdw == 3 ? reinterpret_cast<const void *>(&ndt_build_flat_kernel<3>)
                              : reinterpret_cast<const void *>(&ndt_build_flat_kernel<4>);
    if (dev < 0 || dev >= 16 || !attr_set[sdw == 3 ? 0 : 1][dev]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 16) attr_set[sdw == 3 ? 0 : 1][dev] = true;
    }
    if (sdw == 3)
        hipLaunchKernelGGL((ndt_build_flat_kernel<3>), dim3((unsigned)count), dim3(NDT_FLAT_THREADS), dyn, stream, set,
                           (unsigned)first, (const char *)xyz_dev, (unsigned)n_points, map_stride_bytes, range_limit,
                           range_origins_dev, n_min, eval_factor, s1_shift, s2_shift, hash_log2);
    else
        hipLaunchKernelGGL((ndt_build_flat_kernel<4>), dim3((unsigned)count), dim3(NDT_FLAT_THREADS), dyn, stream, set,
                           (unsigned)first, (const char *)xyz_dev, (unsigned)n_points, map_stride_bytes, range_limit,
                           range_origins_dev, n_min, eval_factor, s1_shift, s2_shift, hash_log2);
    return hipGetLastError();
}
