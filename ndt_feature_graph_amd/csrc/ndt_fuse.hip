// ndt_fuse.hip -- incremental (fused) node maps and their occupancy on CDNA4 (gfx950).
//
// Replaces (reference call sites; perception_oru semantics per SURVEY.md App. A.2-A.3 and DESIGN.md):
//   NDTMap::addPointCloud(origin, cloud, classifierTh, maxz, sensor_noise)   ray-traced insert
//       ndt_feature/src/ndt_feature_src/ndt_feature_fuser_hmt.cpp:92 (0.1, 100, 0.1), :485 (0.06, 25)
//   NDTMap::computeNDTCells(SAMPLE_VARIANCE, 1e5, 255, origin, 0.1)          recursive (N, mean, cov) merge
//       ...fuser_hmt.cpp:94, 486
//   ndt_feature::overlapNDTOccupancyScore(ref, mov, T)                        ndt_feature/include/ndt_feature/ndt_feature_node.h:213-252
//
// One update of B node maps with one cloud each is three launches on one stream:
//   1. ndt_raytrace_kernel   one lane per beam: the cells between sensor and hit (LazyGrid::traceLine: sampled every
//                            `res`, samples rounded to float) receive emptiness evidence.  A cell with a Gaussian gets
//                            the likelihood-weighted log-odds of computeMaximumLikelihoodAlongLine, any other cell
//                            -0.2.  Every update is the float the reference hands to NDTCell::updateOccupancy; they
//                            are summed EXACTLY (64-bit integer atomics in units of 2^-32), so the result does not
//                            depend on the order of the beams.
//   2. ndt_build_kernel<.,1> (csrc/ndt_build.hip, phase A only): the hits are added to the moment accumulators of
//                            their cells (exact integer-valued fp64 atomics).
//   3. ndt_fuse_finalize_kernel  one workgroup per map: occupancy = clamp(occupancy + evidence); per touched cell
//                            computeGaussian (occupancy += n log 1.5; first Gaussian or Chan's pairwise update of
//                            (N, N mean, (N-1) cov) with saturation at maxnumpoints; rescaleCovariance); old Gaussians
//                            whose occupancy fell to <= 0 disappear; all Gaussian cells are ranked in slot order into the
//                            map's OTHER cell array (the old one is read while the new one is written) together with the
//                            rank map the matcher probes.
// Deviation from the reference, restated by the test suite's CPU checker in its `order_free` mode: the reference processes beam
// after beam, so a cell that loses its Gaussian half way through a cloud is treated as empty by the remaining beams,
// and it accumulates in float.  Here every beam sees the cells as they were when the call started.
#include "ndt_math.h"
#include "ndt_wave.h"

#define NDT_FUSE_THREADS 1024
// the finalise kernel: 512 threads = 8 waves = 256 VGPRs each (at 1024 threads the 128-register budget spilled the
// Gaussian update of a cell -- 110 registers, 292 bytes of scratch per lane; 256 threads, three workgroups per CU instead of
// one: add_cloud of 256 maps 3.55 against 3.44 ms, the node builds of --config 4 135.7 against 134.8 ms -- measured, round 4)
#ifndef NDT_FIN2_THREADS
#define NDT_FIN2_THREADS 512
#endif
#define NDT_EMPTY (-1)
#define NDT_DROPPED (-2)   // work-table marker inside ndt_fuse_finalize_kernel: an old Gaussian that this update dropped

namespace {

// One beam through one Gaussian cell: false = the cell is left alone.
NDT_D bool beam_evidence(const NdtCell &c, const double *origin, float ex, float ey, float ez, double sensor_noise, float *upd)
{
#pragma clang fp contract(off)
    sym3 ic;
    {
        const sym3 a = {c.cov[0], c.cov[1], c.cov[2], c.cov[3], c.cov[4], c.cov[5]};
        const double c00 = a.yy * a.zz - a.yz * a.yz;
        const double c01 = a.yz * a.xz - a.xy * a.zz;
        const double c02 = a.xy * a.yz - a.yy * a.xz;
        const double det = a.xx * c00 + a.xy * c01 + a.xz * c02;
        if (det == 0.0 || det != det) return false;
        const double id = 1.0 / det;
        ic.xx = c00 * id; ic.xy = c01 * id; ic.xz = c02 * id;
        ic.yy = (a.xx * a.zz - a.xz * a.xz) * id;
        ic.yz = (a.xy * a.xz - a.xx * a.yz) * id;
        ic.zz = (a.xx * a.yy - a.xy * a.xy) * id;
    }
    const float pox = (float)origin[0], poy = (float)origin[1], poz = (float)origin[2];   // pcl::PointXYZ of the origin
    const d3 v1 = {(double)pox, (double)poy, (double)poz}, v2 = {(double)ex, (double)ey, (double)ez};
    const d3 d = v2 - v1;
    const double nl = sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
    const d3 L = {d.x / nl, d.y / nl, d.z / nl};
    const d3 A = mul(ic, L);
    const d3 Bv = {v2.x - c.mean[0], v2.y - c.mean[1], v2.z - c.mean[2]};
    const double sigma = A.x * L.x + A.y * L.y + A.z * L.z;
    if (sigma == 0) return false;
    const double t = -(A.x * Bv.x + A.y * Bv.y + A.z * Bv.z) / sigma;
    const d3 X = {L.x * t + v2.x, L.y * t + v2.y, L.z * t + v2.z};
    double lik;
    {
        const float xf = (float)X.x, yf = (float)X.y, zf = (float)X.z;            // getLikelihood(pcl::PointXYZ)
        const d3 w = {(double)xf - c.mean[0], (double)yf - c.mean[1], (double)zf - c.mean[2]};
        const double q = dot(w, mul(ic, w));
        lik = (q != q) ? -1.0 : exp(-q / 2);
    }
    const d3 e = {v2.x - origin[0], v2.y - origin[1], v2.z - origin[2]};
    const double l = sqrt(e.x * e.x + e.y * e.y + e.z * e.z);
    const d3 o = {origin[0] - X.x, origin[1] - X.y, origin[2] - X.z};
    const double dist = sqrt(o.x * o.x + o.y * o.y + o.z * o.z);
    if (dist > l) return false;                       // maximum behind the measured end
    const d3 g = {X.x - v2.x, X.y - v2.y, X.z - v2.z};
    const double l2target = sqrt(g.x * g.x + g.y * g.y + g.z * g.z);
    const double sigma_dist = 0.5 * (dist / 30.0);    // distance-dependent sensor noise
    const double snoise = sigma_dist + sensor_noise;
    const double thr = exp(-0.5 * (l2target * l2target) / (snoise * snoise));
    lik *= (1.0 - thr);
    if (lik < 0.3) return false;
    lik = 0.1 * lik + 0.5;                            // evidence for "empty"
    *upd = (float)log((1.0 - lik) / lik);
    return true;
}

// NDTCell::getOccupancyRescaled: 1 - 1 / (1 + exp(occ)) in float arithmetic, expf as the correctly rounded value
NDT_D float occupancy_rescaled(float occ)
{
#pragma clang fp contract(off)
    const float e = (float)exp((double)occ);
    const float o = 1.0f - 1.0f / (1.0f + e);
    return o > 1.0f ? 1.0f : (o < 0.0f ? 0.0f : o);
}

NDT_D unsigned fuse_wave_incl_scan(unsigned v)
{
    const unsigned lane = threadIdx.x & 63u;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        unsigned t = __shfl_up(v, o, 64);
        if (lane >= (unsigned)o) v += t;
    }
    return v;
}

}  // namespace

// Sum of a 32-bit integer over the 64 lanes of a wave, in the vector ALU (DPP row shifts and broadcasts, no LDS
// round trips): inclusive scan within the rows of 16, row totals handed on by the two row broadcasts, total in lane 63.
__device__ __forceinline__ int wave_sum_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);   // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// LazyGrid::traceLine + the occupancy half of NDTMap::addPointCloud.  grid (ceil(n / 256), maps).
extern "C" __global__ __launch_bounds__(256) void ndt_raytrace_kernel(
    NdtSetView set, unsigned first, const char *__restrict__ xyz, unsigned n_points, unsigned stride_bytes,
    size_t map_stride_bytes, const double *__restrict__ origins, double maxz, double sensor_noise)
{
#pragma clang fp contract(off)
    const unsigned map_local = blockIdx.y, map = first + map_local;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const NdtGrid g = set.grid;
    // (no early return: the lanes of a wave add up their updates of a cell before one of them issues the atomic)
    bool beam = i < n_points;
    float ex = 0, ey = 0, ez = 0;
    if (beam) {
        const float *pf = reinterpret_cast<const float *>(xyz + (size_t)map_local * map_stride_bytes + (size_t)i * stride_bytes);
        ex = pf[0]; ey = pf[1]; ez = pf[2];
    }
    if (ex != ex || ey != ey || ez != ez) beam = false;
    const double origin[3] = {origins[map_local * 3], origins[map_local * 3 + 1], origins[map_local * 3 + 2]};
    const double dx = (double)ex - origin[0], dy = (double)ey - origin[1], dz = (double)ez - origin[2];
    const double l = sqrt(dx * dx + dy * dy + dz * dz);
    if (l > 200.0) beam = false;                      // addPointCloud: max_range
    if ((double)ez > maxz) beam = false;              // traceLine: the whole point is dropped
    int N = beam ? (int)(l / g.res) : 0;
    if (N <= 2) N = 0;
    const double sx = dx / (double)(float)N, sy = dy / (double)(float)N, sz = dz / (double)(float)N;
    const double cx = set.centres[map * 3], cy = set.centres[map * 3 + 1], cz = set.centres[map * 3 + 2];
    const uint2 *rankmap = set.rankmap + (size_t)map * ndt_rm_stride(g);
    const NdtCell *cells = ndt_cells_of(set, map, set.cell_sel[map]);
    long long *delta = set.occ_delta + (size_t)map * g.slots;
    unsigned char *touched = set.occ_touched + (size_t)map * (((size_t)g.slots + 255) >> 8);
    int iox = 0, ioy = 0, ioz = 0;                    // idxo = idyo = idzo = 0 like upstream
    const unsigned lane = threadIdx.x & 63u;
    // Cell index of a sample.  The reference rounds the sample to float, p = (float)(origin + f s), and takes
    // floor((p - c) / res + 0.5) + size / 2.  Fast path: w = f (s / res) + ((origin - c) / res + 0.5 + size / 2) -- ONE fp64
    // fma per axis -- differs from the reference's argument by the float rounding of p, at most 2^-24 |p| / res with |p| no
    // larger than the larger of |origin| and |end point| on that axis, plus fp64 rounding (1e-13): a sample whose fraction is
    // within `guard` (twice that bound) of a cell face, odd grid sizes and absurd coordinates take the reference's formula.
    const double inv_res = 1.0 / g.res;
    const double ax = sx * inv_res, ay = sy * inv_res, az = sz * inv_res;
    const double bx = (origin[0] - cx) * inv_res + (0.5 + g.size[0] / 2.0), by = (origin[1] - cy) * inv_res + (0.5 + g.size[1] / 2.0),
                 bz = (origin[2] - cz) * inv_res + (0.5 + g.size[2] / 2.0);
    const bool force_exact = ((g.size[0] | g.size[1] | g.size[2]) & 1) != 0;
    const double pmax = fmax(fmax(fmax(fabs(origin[0]), fabs(origin[1])), fabs(origin[2])),
                             (double)fmaxf(fmaxf(fabsf(ex), fabsf(ey)), fabsf(ez)));
    const double guard = pmax * inv_res * 1.1920928955078125e-7 + 1e-9;           // 2^-23 |p| / res
    const double frac_lim = (force_exact || !(guard < 0.25)) ? -1.0 : 0.5 - guard;
    // (every update of a cell without a Gaussian is the float -0.2f, in units of 2^-32: -13421773 * 2^6, exactly)
    constexpr long long NDT_EMPTY_UPDATE = -858993472ll;
    static_assert((long long)((double)-0.2f * 4294967296.0) == NDT_EMPTY_UPDATE, "the update of a cell seen empty");
    // Neighbouring beams walk through the same cells at the same step (100 k beams per turn: the 64 beams of a wave are
    // 4 cm apart at 10 m).  64 atomics on ONE address in one instruction are served one after the other at the L2
    // (the launch ran at 11 G updates/s); the lanes of a wave therefore add up the updates of a cell -- integers, so
    // the sum is exact and the result the same -- and one lane adds the sum.
    for (int k = 0; ndt_ballot(k < N - 2); k++) {
        int slot = -1;
        long long val = 0;
        if (k < N - 2) {
            const double f = (double)(float)(k + 1);
            const double wx = fma(f, ax, bx), wy = fma(f, ay, by), wz = fma(f, az, bz);
            const double flx = floor(wx), fly = floor(wy), flz = floor(wz);
            int ix = (int)flx, iy = (int)fly, iz = (int)flz;
            const bool nx = !(fabs((wx - flx) - 0.5) <= frac_lim), ny = !(fabs((wy - fly) - 0.5) <= frac_lim),
                       nz = !(fabs((wz - flz) - 0.5) <= frac_lim);
            if (ndt_ballot(nx | ny | nz)) {                 // (rare; a sample far outside the grid is out of bounds on either path)
                if (nx) ix = lazygrid_index((double)(float)(origin[0] + f * sx), cx, g.res, g.size[0]);
                if (ny) iy = lazygrid_index((double)(float)(origin[1] + f * sy), cy, g.res, g.size[1]);
                if (nz) iz = lazygrid_index((double)(float)(origin[2] + f * sz), cz, g.res, g.size[2]);
            }
            if (!(ix == iox && iy == ioy && iz == ioz)) {
                iox = ix; ioy = iy; ioz = iz;
                if ((unsigned)ix < (unsigned)g.size[0] && (unsigned)iy < (unsigned)g.size[1] && (unsigned)iz < (unsigned)g.size[2]) {
                    const int sl = (ix * g.size[1] + iy) * g.size[2] + iz;
                    const int r = ndt_rank_of(rankmap, (unsigned)sl);
                    if (r < 0) {                      // seen empty, no Gaussian to argue with: -0.2f
                        slot = sl;
                        val = NDT_EMPTY_UPDATE;
                    } else {
                        float upd;
                        const NdtCell c = cells[r];
                        if (beam_evidence(c, origin, ex, ey, ez, sensor_noise, &upd)) {
                            slot = sl;
                            val = (long long)((double)upd * 4294967296.0);   // exact: a float below 1 in magnitude times 2^32 is an integer
                        }
                    }
                }
            }
        }
        unsigned long long todo = ndt_ballot(slot >= 0);
        // (Measured and not kept, round 5 -- the launch is bound by the rate of the memory-side atomics, 8-12 G/s, AND by its
        //  ~45 vector instructions per step, each about half of its time: (i) the rank map staged in LDS, 2048-8192 beams per
        //  workgroup: the node builds of --config 4 130-133 ms against 127; (ii) a workgroup-level LDS table of evidence sums
        //  keyed by cell, one global atomic per cell and workgroup: 162 ms -- the compare-and-swap of the leader lane sits in
        //  this loop; (iii) runs of consecutive lanes in the same cell, value x length, ONE atomic instruction for all run heads
        //  instead of this loop: 137 ms -- more atomics in flight at once; without any atomic the same code takes 95 ms;
        //  (iv) a test for "one cell without a Gaussian for every lane that has an update" in front of the loop: 116 against 115 ms.)
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int s0 = __builtin_amdgcn_readlane(slot, leader);        // (the leader is wave-uniform: a register read, no LDS round trip)
            const bool mine = slot == s0;
            const unsigned long long m_mine = ndt_ballot(mine);
            // Most cells a beam crosses hold no Gaussian: every lane's update of the cell is the same -0.2, and the sum is
            // that value times the number of lanes -- scalar arithmetic.  Only when the updates differ (a cell with a
            // Gaussian: every beam has its own evidence) are they summed over the wave: |val| < 2^32, the low 20 bits and
            // the (signed) rest as two 32-bit integers, exactly.  Integer sums either way: the same bits.
            // (Measured and not kept: two samples per trip with both rank-map words requested before either is used --
            //  add_cloud of 256 x 100 k points 4.36 ms against 3.47.)
            const long long v0 = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(val >> 32), leader) << 32) |
                                             (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)val, leader));
            long long v;
            if (ndt_ballot(mine && val != v0) == 0ull) {
                v = v0 * (long long)__popcll(m_mine);
            } else {
                const long long mv = mine ? val : 0ll;
                v = ((long long)wave_sum_i32((int)(mv >> 20)) << 20) + (long long)wave_sum_i32((int)(mv & 0xFFFFF));
            }
            if ((int)lane == leader) {
                atomicAdd(reinterpret_cast<unsigned long long *>(delta + s0), (unsigned long long)v);
                touched[(unsigned)s0 >> 8] = 1;      // (a plain byte store: everybody writes the same 1)
            }
            todo &= ~m_mine;
        }
    }
}

// computeNDTCells of an incremental update: one workgroup per map.
extern "C" __global__ __launch_bounds__(NDT_FIN2_THREADS) void ndt_fuse_finalize_kernel(
    NdtSetView set, unsigned first, unsigned n_points, int n_min, double eval_factor, double maxnumpoints,
    float occupancy_limit, int s1_shift, int s2_shift)
{
    __shared__ unsigned s_wave_cnt[NDT_FIN2_THREADS / 64];
    __shared__ unsigned s_binned;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    constexpr unsigned nthreads = NDT_FIN2_THREADS, nwaves = NDT_FIN2_THREADS / 64;
    const unsigned map = first + blockIdx.x;
    const NdtGrid g = set.grid;
    const uint32_t cap = g.max_cells;
    int32_t *wtable = set.wtable + (size_t)map * g.slots;
    uint2 *rankmap = set.rankmap + (size_t)map * ndt_rm_stride(g);
    uint32_t *bitmap = set.bitmap + (size_t)map * ((g.slots + 31) >> 5);
    const unsigned bm_words = (unsigned)((g.slots + 31) >> 5);
    NdtAcc *acc = set.acc + (size_t)map * cap;
    const uint32_t *acc_slot = set.acc_slot + (size_t)map * cap;
    NdtMapCounters *ctr = set.counters + map;
    const uint32_t sel = set.cell_sel[map];
    const NdtCell *cells_old = ndt_cells_of(set, map, sel);
    NdtCell *cells_new = ndt_cells_of(set, map, sel ^ 1u);
    float *occ = set.occ + (size_t)map * g.slots;
    long long *delta = set.occ_delta + (size_t)map * g.slots;
    const double cx = set.centres[map * 3 + 0], cy = set.centres[map * 3 + 1], cz = set.centres[map * 3 + 2];
    const double res = g.res;
    const double hx = g.size[0] / 2.0, hy = g.size[1] / 2.0, hz = g.size[2] / 2.0;
    if (tid == 0) s_binned = 0;

    // ---- 1. the beams' evidence: occupancy = clamp(occupancy + sum of the updates) ---------------------------------
    // Only the blocks of 256 slots that a beam marked (ndt_raytrace_kernel): a scan leaves evidence in a few per cent of a node
    // map's slots, and reading all of them was 640 KB per map and scan.  64 block flags per load; the k-th marked block of a
    // group goes to wave k mod nwaves.
    {
        const unsigned nblk = ((unsigned)g.slots + 255u) >> 8;
        unsigned char *touched = set.occ_touched + (size_t)map * nblk;
        for (unsigned b0 = 0; b0 < nblk; b0 += 64u) {
            const unsigned bi = b0 + lane;
            unsigned long long marked = ndt_ballot(bi < nblk && touched[bi] != 0);
            for (unsigned k = 0; marked; k++) {
                const unsigned b = b0 + (unsigned)__builtin_ctzll(marked);
                marked &= marked - 1ull;
                if (k % nwaves != wave) continue;
                const unsigned s_end = min((unsigned)g.slots, (b + 1u) << 8);
                for (unsigned s = (b << 8) + lane; s < s_end; s += 64u) {
                    const long long d = delta[s];
                    if (d != 0) {
                        float o = (float)((double)occ[s] + (double)d * (1.0 / 4294967296.0));
                        o = o > occupancy_limit ? occupancy_limit : (o < -occupancy_limit ? -occupancy_limit : o);
                        occ[s] = o;
                        delta[s] = 0;
                    }
                }
            }
        }
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // accumulator atomics bypass the L1
        __syncthreads();                                                   // (every wave has read the flags)
        for (unsigned i = tid; i < nblk; i += nthreads)
            if (touched[i] != 0) touched[i] = 0;
    }

    // ---- 2. computeGaussian of every cell that received points --------------------------------------------------
    unsigned n_alloc = ctr->n_alloc;
    if (n_alloc > cap) n_alloc = cap;
    const unsigned n_old = ctr->n_cells > cap ? cap : ctr->n_cells;
    const double IS1 = ldexp(1.0, -s1_shift), IS2 = ldexp(1.0, -s2_shift);
    unsigned binned = 0;
    for (unsigned id = tid; id < n_alloc; id += nthreads) {
        const NdtAcc a = acc[id];
        NdtCell c;
        c.n = 0; c.slot = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) c.mean[k] = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) c.cov[k] = 0;
        const unsigned long long n = (unsigned long long)a.n;
        binned += (unsigned)n;
        if (n > 0) {
            const unsigned slot = acc_slot[id];
            const int old_rank = ndt_rank_of(rankmap, slot);        // (the rank map still describes the old cells)
            const bool had = old_rank >= 0 && occ[slot] > 0.0f;      // a Gaussian the beams did not take away
            // occupancy += n log(0.6 / 0.4), clamped
            float o = occ[slot] + (float)((double)n * NDT_LOGODD_OCC);
            o = o > occupancy_limit ? occupancy_limit : (o < -occupancy_limit ? -occupancy_limit : o);
            occ[slot] = o;
            const int iz = slot % g.size[2], iy = (slot / g.size[2]) % g.size[1], ix = slot / (g.size[2] * g.size[1]);
            const double ox = cx + (ix - hx) * res, oy = cy + (iy - hy) * res, oz = cz + (iz - hz) * res;   // cell origin
            const double dn = (double)n;
            double t2[3], S[6];   // sum of the new points' offsets from the cell origin [m], sum of their outer products [m^2]
#pragma unroll
            for (int k = 0; k < 3; k++) t2[k] = (double)a.s1[k] * IS1 * res;
#pragma unroll
            for (int k = 0; k < 6; k++) S[k] = (double)a.s2[k] * IS2 * (res * res);
            const double m2[3] = {t2[0] / dn, t2[1] / dn, t2[2] / dn};
            // c2 = sum (d - m2)(d - m2)^T
            double c2[6] = {S[0] - dn * m2[0] * m2[0], S[1] - dn * m2[0] * m2[1], S[2] - dn * m2[0] * m2[2],
                            S[3] - dn * m2[1] * m2[1], S[4] - dn * m2[1] * m2[2], S[5] - dn * m2[2] * m2[2]};
            bool gauss = false;
            double mean[3] = {0, 0, 0}, C6[6] = {0, 0, 0, 0, 0, 0};
            double Nn = 0;
            if (o > 0.0f) {
                if (had) {
                    // Chan's pairwise update of (N, N mean, (N - 1) cov), offsets from the cell origin
                    const NdtCell oc = cells_old[old_rank];
                    double N = (double)oc.n;
                    const double mu[3] = {oc.mean[0] - ox, oc.mean[1] - oy, oc.mean[2] - oz};
                    double ms[3] = {mu[0] * N, mu[1] * N, mu[2] * N};
                    double cs[6];
#pragma unroll
                    for (int k = 0; k < 6; k++) cs[k] = oc.cov[k] * (N - 1.0);
                    const double w1 = N / (dn * (N + dn)), w2 = dn / N;
                    const double c3[3] = {ms[0] * w2 - t2[0], ms[1] * w2 - t2[1], ms[2] * w2 - t2[2]};
                    cs[0] += c2[0] + w1 * (c3[0] * c3[0]); cs[1] += c2[1] + w1 * (c3[0] * c3[1]); cs[2] += c2[2] + w1 * (c3[0] * c3[2]);
                    cs[3] += c2[3] + w1 * (c3[1] * c3[1]); cs[4] += c2[4] + w1 * (c3[1] * c3[2]); cs[5] += c2[5] + w1 * (c3[2] * c3[2]);
#pragma unroll
                    for (int k = 0; k < 3; k++) ms[k] += t2[k];
                    N += dn;
                    if (maxnumpoints > 0 && maxnumpoints < N) {     // "sliding average"
#pragma unroll
                        for (int k = 0; k < 3; k++) ms[k] *= maxnumpoints / N;
#pragma unroll
                        for (int k = 0; k < 6; k++) cs[k] *= (maxnumpoints - 1.0) / (N - 1.0);
                        N = maxnumpoints;
                    }
#pragma unroll
                    for (int k = 0; k < 3; k++) mean[k] = ms[k] / N;
#pragma unroll
                    for (int k = 0; k < 6; k++) C6[k] = cs[k] / (N - 1.0);
                    Nn = N;
                    gauss = true;
                } else if (n >= 2 && n >= (unsigned long long)n_min) {
#pragma unroll
                    for (int k = 0; k < 3; k++) mean[k] = m2[k];
#pragma unroll
                    for (int k = 0; k < 6; k++) C6[k] = c2[k] / (dn - 1.0);
                    Nn = dn;
                    gauss = true;
                }
            }
            if (gauss) {
                // NDTCell::rescaleCovariance
                double E[3][3] = {{C6[0], C6[1], C6[2]}, {C6[1], C6[3], C6[4]}, {C6[2], C6[4], C6[5]}}, V[3][3];
                jacobi_static<3, true>(E, V);
                double ev[3] = {E[0][0], E[1][1], E[2][2]};
                const double mx = dmax3(ev[0], ev[1], ev[2]), mn = dmin3(ev[0], ev[1], ev[2]);
                if (mx > 0 && mn > NDT_DEGENERATE_REL * mx) {
                    bool recalc = false;
#pragma unroll
                    for (int k = 0; k < 3; k++)
                        if (mx > ev[k] * eval_factor) { ev[k] = mx / eval_factor; recalc = true; }
                    if (recalc) {
                        double R[3][3];
#pragma unroll
                        for (int r = 0; r < 3; r++)
#pragma unroll
                            for (int q2 = r; q2 < 3; q2++) {
                                double s = 0;
#pragma unroll
                                for (int k = 0; k < 3; k++) s += V[r][k] * ev[k] * V[q2][k];
                                R[r][q2] = s;
                            }
                        C6[0] = R[0][0]; C6[1] = R[0][1]; C6[2] = R[0][2]; C6[3] = R[1][1]; C6[4] = R[1][2]; C6[5] = R[2][2];
                    }
                    c.mean[0] = ox + mean[0]; c.mean[1] = oy + mean[1]; c.mean[2] = oz + mean[2];
#pragma unroll
                    for (int k = 0; k < 6; k++) c.cov[k] = C6[k];
                    c.n = (uint32_t)Nn;
                    c.slot = slot;
                }
            }
            if (c.n == 0) {
                // touched, no Gaussian: the cell leaves the bitmap and its slot the work table; an old Gaussian that is
                // dropped here keeps a marker there until step 3 has passed it
                __hip_atomic_fetch_and(&bitmap[slot >> 5], ~(1u << (slot & 31u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                wtable[slot] = old_rank >= 0 ? NDT_DROPPED : NDT_EMPTY;
            }
        }
        *reinterpret_cast<NdtCell *>(acc + id) = c;    // the record waits in its own accumulator
    }
    if (binned) atomicAdd(&s_binned, binned);
    __syncthreads();   // the decisions above (work table, rank table, occupancy) are read below by other threads

    // ---- 3. Gaussians that received no points: they stay while their occupancy is positive -------------------------
    for (unsigned r = tid; r < n_old; r += nthreads) {
        const unsigned slot = cells_old[r].slot;
        const int wt = wtable[slot];                   // (a stale EMPTY cannot be read: this workgroup's L1 was
                                                       // invalidated after the accumulation)
        if (wt == NDT_DROPPED) { wtable[slot] = NDT_EMPTY; continue; }   // touched and dropped above
        if (wt != NDT_EMPTY) continue;                 // touched: decided above
        if (occ[slot] > 0.0f) __hip_atomic_fetch_or(&bitmap[slot >> 5], 1u << (slot & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // bitmap atomics are performed at the memory side
    __syncthreads();

    // ---- 4. rank every Gaussian cell in slot order into the other cell array -------------------------------------------
    const unsigned words_per_wave = (bm_words + nwaves - 1) / nwaves;
    const unsigned wb = min(bm_words, wave * words_per_wave), we = min(bm_words, wb + words_per_wave);
    const bool ovf = __hip_atomic_load(&ctr->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    auto valid_bits = [&](unsigned w, unsigned bits) {   // capacity overflow: slots whose id is past the capacity
        unsigned vmask = 0;
        for (unsigned b = bits; b; b &= b - 1) {
            const int bit = __ffs((int)b) - 1;
            const int id = wtable[w * 32 + bit];
            if (id == NDT_EMPTY || ((uint32_t)id < n_alloc && reinterpret_cast<const NdtCell *>(acc + id)->n > 0)) vmask |= 1u << bit;
        }
        return vmask;
    };
    {
        unsigned cnt = 0;
        for (unsigned w = wb + lane; w < we; w += 64u) {
            const unsigned bits = bitmap[w];
            cnt += (unsigned)__popc((ovf && bits) ? valid_bits(w, bits) : bits);
        }
        const unsigned incl = fuse_wave_incl_scan(cnt);
        if (lane == 63) s_wave_cnt[wave] = incl;
    }
    __syncthreads();
    unsigned running = 0, total_cells = 0;
    for (unsigned k = 0; k < nwaves; k++) {
        const unsigned c2 = s_wave_cnt[k];
        if (k < wave) running += c2;
        total_cells += c2;
    }
    for (unsigned step = wb; step < we; step += 64u) {
        const unsigned w = step + lane;
        const unsigned bits = (w < we) ? bitmap[w] : 0u;
        const uint2 old = (w < we) ? rankmap[w] : make_uint2(0u, 0u);   // the old cells of this word: bits and first rank
        if (!ndt_ballot((bits | old.x) != 0u)) continue;
        const unsigned vmask = (ovf && bits) ? valid_bits(w, bits) : bits;
        const unsigned cnt = (unsigned)__popc(vmask);
        const unsigned incl = fuse_wave_incl_scan(cnt);
        unsigned before = running + incl - cnt;
        running += __shfl(incl, 63, 64);
        if (vmask | old.x) rankmap[w] = make_uint2(vmask, before);
        for (unsigned b = bits; b; b &= b - 1u) {
            const int bit = __ffs((int)b) - 1;
            const unsigned slot = w * 32u + (unsigned)bit;
            const int id = wtable[slot];
            if (vmask & (1u << bit)) {
                NdtCell c = (id == NDT_EMPTY) ? cells_old[ndt_rank_in_word(old, (unsigned)bit)] : *reinterpret_cast<const NdtCell *>(acc + id);
                c.slot = slot;
                if (before < cap) cells_new[before] = c;
                before++;
            }
            wtable[slot] = NDT_EMPTY;
        }
        if (bits) bitmap[w] = 0u;
    }
    __syncthreads();

    // ---- 5. scratch back to its clean state, counters ---------------------------------------------------------------------
    {
        unsigned long long *z = reinterpret_cast<unsigned long long *>(acc);
        for (unsigned k = tid; k < n_alloc * 10u; k += nthreads) z[k] = 0ull;
    }
    if (tid == 0) {
        if (total_cells > cap) { total_cells = cap; ctr->overflow = 1u; }
        ctr->n_cells = total_cells;
        ctr->n_alloc = 0;
        ctr->n_dropped = n_points - s_binned;
        set.cell_sel[map] = sel ^ 1u;
    }
}

// ndt_feature::overlapNDTOccupancyScore for n links: one workgroup per link, the cells of `mov` are dealt to its
// threads; counts are integers, the squared differences are added in a fixed order (thread-local in slot order, then a
// fixed tree): run-to-run identical.
extern "C" __global__ __launch_bounds__(256) void ndt_overlap_kernel(
    NdtSetView rset, const uint32_t *__restrict__ ridx, NdtSetView mset, const uint32_t *__restrict__ midx,
    const double *__restrict__ T16, double *__restrict__ score, long long *__restrict__ nb_sum)
{
#pragma clang fp contract(off)
    __shared__ double s_sum[256];
    __shared__ unsigned s_cnt[256];
    const unsigned link = blockIdx.x, tid = threadIdx.x;
    const unsigned rm = ridx[link], mm = midx[link];
    const NdtGrid gr = rset.grid, gm = mset.grid;
    const float *occ_r = rset.occ + (size_t)rm * gr.slots;
    const float *occ_m = mset.occ + (size_t)mm * gm.slots;
    const double *T = T16 + (size_t)link * 16;
    const double mcx = mset.centres[mm * 3], mcy = mset.centres[mm * 3 + 1], mcz = mset.centres[mm * 3 + 2];
    const double rcx = rset.centres[rm * 3], rcy = rset.centres[rm * 3 + 1], rcz = rset.centres[rm * 3 + 2];
    double sum = 0.0;
    unsigned cnt = 0;
    for (unsigned s = tid; s < (unsigned)gm.slots; s += 256u) {
        const float om = occ_m[s];
        if (om == 0.0f) continue;                                  // rescaled occupancy exactly 0.5: no reading
        const double mov_occ = (double)occupancy_rescaled(om);
        if (mov_occ == 0.5) continue;
        const int iz = s % gm.size[2], iy = (s / gm.size[2]) % gm.size[1], ix = s / (gm.size[2] * gm.size[1]);
        // NDTCell::getCenter(): float
        const float cfx = (float)(mcx + ((double)ix - (double)(gm.size[0] / 2)) * gm.res);
        const float cfy = (float)(mcy + ((double)iy - (double)(gm.size[1] / 2)) * gm.res);
        const float cfz = (float)(mcz + ((double)iz - (double)(gm.size[2] / 2)) * gm.res);
        const double e0 = cfx, e1 = cfy, e2 = cfz;
        const float px = (float)(T[0] * e0 + T[4] * e1 + T[8] * e2 + T[12]);
        const float py = (float)(T[1] * e0 + T[5] * e1 + T[9] * e2 + T[13]);
        const float pz = (float)(T[2] * e0 + T[6] * e1 + T[10] * e2 + T[14]);
        const int jx = lazygrid_index((double)px, rcx, gr.res, gr.size[0]);
        const int jy = lazygrid_index((double)py, rcy, gr.res, gr.size[1]);
        const int jz = lazygrid_index((double)pz, rcz, gr.res, gr.size[2]);
        if ((unsigned)jx >= (unsigned)gr.size[0] || (unsigned)jy >= (unsigned)gr.size[1] || (unsigned)jz >= (unsigned)gr.size[2]) continue;
        const float orf = occ_r[(jx * gr.size[1] + jy) * gr.size[2] + jz];
        const double ref_occ = (double)occupancy_rescaled(orf);
        if (ref_occ != 0.5) {
            cnt++;
            const double diff = mov_occ - ref_occ;
            sum += diff * diff;
        }
    }
    s_sum[tid] = sum;
    s_cnt[tid] = cnt;
    __syncthreads();
    for (unsigned o = 128; o > 0; o >>= 1) {
        if (tid < o) { s_sum[tid] += s_sum[tid + o]; s_cnt[tid] += s_cnt[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        nb_sum[link] = (long long)s_cnt[0];
        score[link] = s_cnt[0] ? s_sum[0] / (1. * (double)s_cnt[0]) : 1.;
    }
}

// The same score from LISTS of the moving maps' cells with a reading ((slot, occupancy) pairs in slot order: ndt_occ_list_kernel):
// a fused node map has readings in 2-9 % of its slots, and every link of a replay read the whole dense array of its moving
// map -- 320 KB per link, 6.4 GB for the 19 900 links of 200 nodes -- to find them.  list_of_link[link]: which list; the
// pairs of list u are pairs[offs[u] .. offs[u + 1]).  Thread t takes entries t, t + 256, ... of the list (thread-local sums
// in list order, then the fixed tree): run-to-run identical, and equal to the dense kernel's score up to the order of the sum.
extern "C" __global__ __launch_bounds__(256) void ndt_overlap_lists_kernel(
    NdtSetView rset, const uint32_t *__restrict__ ridx, NdtSetView mset, const uint32_t *__restrict__ midx,
    const uint32_t *__restrict__ list_of_link, const unsigned *__restrict__ offs, const uint2 *__restrict__ pairs,
    const double *__restrict__ T16, double *__restrict__ score, long long *__restrict__ nb_sum)
{
#pragma clang fp contract(off)
    __shared__ double s_sum[256];
    __shared__ unsigned s_cnt[256];
    const unsigned link = blockIdx.x, tid = threadIdx.x;
    const unsigned rm = ridx[link], mm = midx[link], u = list_of_link[link];
    const NdtGrid gr = rset.grid, gm = mset.grid;
    const float *occ_r = rset.occ + (size_t)rm * gr.slots;
    const double *T = T16 + (size_t)link * 16;
    const double mcx = mset.centres[mm * 3], mcy = mset.centres[mm * 3 + 1], mcz = mset.centres[mm * 3 + 2];
    const double rcx = rset.centres[rm * 3], rcy = rset.centres[rm * 3 + 1], rcz = rset.centres[rm * 3 + 2];
    double sum = 0.0;
    unsigned cnt = 0;
    const unsigned e1 = offs[u + 1u];
    for (unsigned e = offs[u] + tid; e < e1; e += 256u) {
        const uint2 pr = pairs[e];
        const unsigned s = pr.x;
        const double mov_occ = (double)occupancy_rescaled(__uint_as_float(pr.y));
        if (mov_occ == 0.5) continue;
        const int iz = s % gm.size[2], iy = (s / gm.size[2]) % gm.size[1], ix = s / (gm.size[2] * gm.size[1]);
        // NDTCell::getCenter(): float
        const float cfx = (float)(mcx + ((double)ix - (double)(gm.size[0] / 2)) * gm.res);
        const float cfy = (float)(mcy + ((double)iy - (double)(gm.size[1] / 2)) * gm.res);
        const float cfz = (float)(mcz + ((double)iz - (double)(gm.size[2] / 2)) * gm.res);
        const double e0 = cfx, e1_ = cfy, e2 = cfz;
        const float px = (float)(T[0] * e0 + T[4] * e1_ + T[8] * e2 + T[12]);
        const float py = (float)(T[1] * e0 + T[5] * e1_ + T[9] * e2 + T[13]);
        const float pz = (float)(T[2] * e0 + T[6] * e1_ + T[10] * e2 + T[14]);
        const int jx = lazygrid_index((double)px, rcx, gr.res, gr.size[0]);
        const int jy = lazygrid_index((double)py, rcy, gr.res, gr.size[1]);
        const int jz = lazygrid_index((double)pz, rcz, gr.res, gr.size[2]);
        if ((unsigned)jx >= (unsigned)gr.size[0] || (unsigned)jy >= (unsigned)gr.size[1] || (unsigned)jz >= (unsigned)gr.size[2]) continue;
        const float orf = occ_r[(jx * gr.size[1] + jy) * gr.size[2] + jz];
        const double ref_occ = (double)occupancy_rescaled(orf);
        if (ref_occ != 0.5) {
            cnt++;
            const double diff = mov_occ - ref_occ;
            sum += diff * diff;
        }
    }
    s_sum[tid] = sum;
    s_cnt[tid] = cnt;
    __syncthreads();
    for (unsigned o = 128; o > 0; o >>= 1) {
        if (tid < o) { s_sum[tid] += s_sum[tid + o]; s_cnt[tid] += s_cnt[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        nb_sum[link] = (long long)s_cnt[0];
        score[link] = s_cnt[0] ? s_sum[0] / (1. * (double)s_cnt[0]) : 1.;
    }
}

// ndt_feature::discardCell(map, pt) (utils.h:229-236; fuser_hmt.cpp:229-232): the cells that hold the given points lose
// their Gaussian.  One workgroup; the surviving cells are compacted in place in rank order (a cell never moves up, and a
// chunk of 1024 records is read completely before any of it is rewritten), then the first ranks of the rank map's words
// are rewritten (its bits were cleared while marking).
extern "C" __global__ __launch_bounds__(NDT_FUSE_THREADS) void ndt_discard_kernel(NdtSetView set, unsigned map,
                                                                                  const float *__restrict__ xyz, unsigned n_pts)
{
    __shared__ unsigned s_wave[NDT_FUSE_THREADS / 64];
    __shared__ unsigned s_base;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const NdtGrid g = set.grid;
    uint2 *rankmap = set.rankmap + (size_t)map * ndt_rm_stride(g);
    NdtCell *cells = ndt_cells_of(set, map, set.cell_sel ? set.cell_sel[map] : 0u);
    NdtMapCounters *ctr = set.counters + map;
    const unsigned n_old = ctr->n_cells > g.max_cells ? g.max_cells : ctr->n_cells;
    const double cx = set.centres[map * 3], cy = set.centres[map * 3 + 1], cz = set.centres[map * 3 + 2];
    // 1. mark: the cells that hold a point (NDTMap::getCellAtPoint) leave the rank map's bits
    for (unsigned i = tid; i < n_pts; i += NDT_FUSE_THREADS) {
        const int ix = lazygrid_index((double)xyz[3 * i], cx, g.res, g.size[0]);
        const int iy = lazygrid_index((double)xyz[3 * i + 1], cy, g.res, g.size[1]);
        const int iz = lazygrid_index((double)xyz[3 * i + 2], cz, g.res, g.size[2]);
        if ((unsigned)ix >= (unsigned)g.size[0] || (unsigned)iy >= (unsigned)g.size[1] || (unsigned)iz >= (unsigned)g.size[2]) continue;
        const int slot = (ix * g.size[1] + iy) * g.size[2] + iz;
        atomicAnd(&rankmap[slot >> 5].x, ~(1u << (slot & 31)));
    }
    if (tid == 0) s_base = 0;
    __syncthreads();
    // 2. compact in rank order, 1024 records at a time
    for (unsigned c0 = 0; c0 < n_old; c0 += NDT_FUSE_THREADS) {
        const unsigned r = c0 + tid;
        NdtCell c;
        bool keep = false;
        if (r < n_old) {
            c = cells[r];
            keep = ((__hip_atomic_load(&rankmap[c.slot >> 5].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (c.slot & 31u)) & 1u) != 0u;
        }
        const unsigned long long m = ndt_ballot(keep);
        const unsigned before = (unsigned)__popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64u - lane))));
        if (lane == 0) s_wave[wave] = (unsigned)__popcll(m);
        __syncthreads();
        unsigned off = s_base;
        for (unsigned k = 0; k < wave; k++) off += s_wave[k];
        if (keep) {
            cells[off + before] = c;
        }
        __syncthreads();
        if (tid == 0) {
            unsigned t = 0;
            for (unsigned k = 0; k < NDT_FUSE_THREADS / 64; k++) t += s_wave[k];
            s_base += t;
        }
        __syncthreads();
    }
    const unsigned n_new = s_base;
    // 3. rank map: the bits are those that survived step 1; the rank of the first cell of every 32-slot word
    for (unsigned r = tid; r < n_new; r += NDT_FUSE_THREADS) {
        const unsigned slot = cells[r].slot;
        if (r == 0 || (cells[r - 1].slot >> 5) != (slot >> 5)) rankmap[slot >> 5].y = r;
    }
    if (tid == 0) ctr->n_cells = n_new;
}

hipError_t ndt_launch_discard(const NdtSetView &set, size_t map, const float *xyz_dev, size_t n_pts, hipStream_t stream)
{
    hipLaunchKernelGGL(ndt_discard_kernel, dim3(1), dim3(NDT_FUSE_THREADS), 0, stream, set, (unsigned)map, xyz_dev, (unsigned)n_pts);
    return hipGetLastError();
}

hipError_t ndt_launch_fuse(const NdtSetView &set, size_t first, size_t count, const void *xyz_dev, size_t n_points,
                           size_t stride_bytes, size_t map_stride_bytes, const double *origins_dev,
                           const NdtFuseParams &prm, int nice, hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    if (n_points) {
        // One wave per workgroup: the waves of a launch are independent (no LDS, no barrier), rays differ a lot in length,
        // and a 256-thread workgroup held its four wave slots until its longest ray was done.  Measured, round 4: 256 / 128 /
        // 64 threads: add_cloud of 256 x 100 k points 3.45 / 3.43 / 3.37 ms, the 5000 node builds of `bench.py --config 4`
        // 135.3 / 132.7 / 127.1 ms (294 -> 310 k gated edge registrations/s).  The same rays in the same waves: the same bits.
        const char *te = getenv("NDTGPU_RAY_THREADS");           // (experiments: 64 / 128 / 256 threads per workgroup)
        const unsigned rt = te && (atoi(te) == 128 || atoi(te) == 256) ? (unsigned)atoi(te) : 64u;
        const unsigned blocks = (unsigned)((n_points + rt - 1) / rt);
        hipLaunchKernelGGL(ndt_raytrace_kernel, dim3(blocks, (unsigned)count), dim3(rt), 0, stream, set, (unsigned)first,
                           (const char *)xyz_dev, (unsigned)n_points, (unsigned)stride_bytes, map_stride_bytes, origins_dev,
                           prm.maxz, prm.sensor_noise);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    int s1 = 0, s2 = 0;
    // the hits: NaN dropped, |p - origin| > 200 dropped (addPointCloud's max_range), z > maxz dropped, outside the grid dropped
    hipError_t e = ndt_launch_accumulate(set, first, count, xyz_dev, n_points, stride_bytes, map_stride_bytes, 200.0,
                                         origins_dev, prm.maxz, nice, &s1, &s2, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ndt_fuse_finalize_kernel, dim3((unsigned)count), dim3(NDT_FIN2_THREADS), 0, stream, set,
                       (unsigned)first, (unsigned)n_points, prm.n_min, prm.eval_factor, prm.maxnumpoints,
                       (float)prm.occupancy_limit, s1, s2);
    return hipGetLastError();
}

hipError_t ndt_launch_overlap_lists(const NdtSetView &rset, const uint32_t *ridx_dev, const NdtSetView &mset,
                                    const uint32_t *midx_dev, const uint32_t *list_of_link_dev, const unsigned *offs_dev,
                                    const void *pairs_dev, const double *T16_dev, size_t n_links, double *score_dev,
                                    long long *nb_dev, hipStream_t stream)
{
    if (n_links == 0) return hipSuccess;
    hipLaunchKernelGGL(ndt_overlap_lists_kernel, dim3((unsigned)n_links), dim3(256), 0, stream, rset, ridx_dev, mset, midx_dev,
                       list_of_link_dev, offs_dev, (const uint2 *)pairs_dev, T16_dev, score_dev, nb_dev);
    return hipGetLastError();
}

hipError_t ndt_launch_overlap(const NdtSetView &rset, const uint32_t *ridx_dev, const NdtSetView &mset,
                              const uint32_t *midx_dev, const double *T16_dev, size_t n_links, double *score_dev,
                              long long *nb_dev, hipStream_t stream)
{
    if (n_links == 0) return hipSuccess;
    hipLaunchKernelGGL(ndt_overlap_kernel, dim3((unsigned)n_links), dim3(256), 0, stream, rset, ridx_dev, mset, midx_dev,
                       T16_dev, score_dev, nb_dev);
    return hipGetLastError();
}
