// ndt_build.hip -- NDT grid build on CDNA4 (gfx950): K1 voxel keying, K2 per-cell moment
// accumulation, K3 Gaussian finalisation, fused in ONE kernel with one workgroup per map.
//
// Replaces (reference call sites, perception_oru semantics per SURVEY.md App. A):
//   LazyGrid::getIndexForPoint / addPoint, NDTMap::loadPointCloud(cloud, range)
//       ndt_feature/src/ndt_feature_src/ndt_feature_fuser_hmt.cpp:195-226
//   NDTMap::computeNDTCells(CELL_UPDATE_MODE_SAMPLE_VARIANCE) -> NDTCell::computeGaussian +
//   rescaleCovariance                                      ...fuser_hmt.cpp:227, ndt_odom_debug.cpp:179
//
// Design (DESIGN.md "Build kernel"):
//   * one 1024-thread workgroup owns one map: its slot table, accumulators and cells are touched by
//     this workgroup only, so every atomic is workgroup scope and nothing crosses XCDs; a batch of
//     B scans is B workgroups (B >= 256 fills the chip; a single scan is latency-, not bandwidth-bound
//     and is not the metric's case).
//   * raw scan read once, coalesced; per point: key, range/NaN filter, offset from its cell centre in
//     units of the cell size, converted to FIXED POINT so that the per-cell sums are integer sums:
//     exactly associative -> bit-identical results whatever the order of the atomics.
//   * wavefront pre-reduction: lanes of a wave that hit the same cell (the normal case for an
//     angularly ordered laser sweep) are summed with shuffles and ONE lane issues the 10 atomics.
//   * finalisation in the same launch: mean / sample covariance from the integer moments, 3x3 Jacobi
//     eigen-decomposition, eigenvalue floor, then cells are ranked in slot order by a block scan over
//     the dense table (deterministic cell order), and the scratch is left zeroed for the next build.
//   HBM algorithmic bytes per scan: 12*N (points) + 80*M (cell records)  (SURVEY.md 8d).
#include "ndt_math.h"

#define NDT_BUILD_THREADS 1024
#define NDT_EMPTY (-1)

extern "C" __global__ void ndt_build_kernel(NdtSetView set, unsigned first, const char *__restrict__ xyz,
                                            unsigned n_points, unsigned stride_bytes, size_t map_stride_bytes,
                                            double range_limit, const double *__restrict__ range_origins, int n_min,
                                            double eval_factor, int s2_shift);

namespace {

NDT_D long long wave_sum(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

NDT_D unsigned long long lanemask_lt()
{
    unsigned lane = threadIdx.x & 63u;
    return (lane == 0) ? 0ull : (~0ull >> (64u - lane));
}

// slot -> accumulator id, allocating on first touch.  Lock-free: a racing loser wastes one id
// (left with n == 0, skipped by the finaliser).
NDT_D int get_or_assign(int32_t *table, int slot, uint32_t *acc_slot, NdtMapCounters *ctr, uint32_t cap)
{
    int id = table[slot];   // may be a stale EMPTY from L1; a non-EMPTY value is always final
    if (id != NDT_EMPTY) return id;
    unsigned nid = __hip_atomic_fetch_add(&ctr->n_alloc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    int expected = NDT_EMPTY;
    if (__hip_atomic_compare_exchange_strong(&table[slot], &expected, (int)nid, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_WORKGROUP)) {
        if (nid < cap) acc_slot[nid] = (uint32_t)slot;
        else __hip_atomic_store(&ctr->overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (int)nid;
    }
    return expected;   // somebody else assigned it first
}

NDT_D void atomic_add_ll(long long *p, long long v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace

extern "C" __global__ __launch_bounds__(NDT_BUILD_THREADS) void ndt_build_kernel(
    NdtSetView set, unsigned first, const char *__restrict__ xyz, unsigned n_points, unsigned stride_bytes,
    size_t map_stride_bytes, double range_limit, const double *__restrict__ range_origins, int n_min,
    double eval_factor, int s2_shift)
{
    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63u;
    const unsigned map = first + blockIdx.x;
    const NdtGrid g = set.grid;
    const uint32_t cap = g.max_cells;
    int32_t *table = set.table + (size_t)map * g.slots;
    NdtAcc *acc = set.acc + (size_t)map * cap;
    NdtCell *cells = set.cells + (size_t)map * cap;
    uint32_t *acc_slot = set.acc_slot + (size_t)map * cap;
    NdtMapCounters *ctr = set.counters + map;
    const double cx = set.centres[map * 3 + 0], cy = set.centres[map * 3 + 1], cz = set.centres[map * 3 + 2];
    const double res = g.res;
    double ox = 0, oy = 0, oz = 0;
    if (range_origins) { ox = range_origins[blockIdx.x * 3]; oy = range_origins[blockIdx.x * 3 + 1]; oz = range_origins[blockIdx.x * 3 + 2]; }
    const char *pts = xyz + (size_t)blockIdx.x * map_stride_bytes;
    const double S1 = (double)(1ull << NDT_S1_SHIFT), IS1 = 1.0 / S1;
    const double S2 = (double)(1ull << s2_shift);

    __shared__ unsigned s_wave_cnt[NDT_BUILD_THREADS / 64];
    __shared__ unsigned s_base;
    __shared__ unsigned s_dropped;
    if (tid == 0) { s_base = 0; s_dropped = 0; }
    __syncthreads();

    // ---------------- phase A: key + accumulate --------------------------------------------
    unsigned dropped = 0;
    for (unsigned base = 0; base < n_points; base += NDT_BUILD_THREADS) {
        unsigned i = base + tid;
        int slot = -1;
        long long q[9];
#pragma unroll
        for (int k = 0; k < 9; k++) q[k] = 0;
        if (i < n_points) {
            const float *pf = (const float *)(pts + (size_t)i * stride_bytes);
            double px = (double)pf[0], py = (double)pf[1], pz = (double)pf[2];
            bool ok = !(isnan(px) || isnan(py) || isnan(pz));
            if (ok && range_limit > 0) {
#pragma clang fp contract(off)
                double dx = px - ox, dy = py - oy, dz = pz - oz;
                ok = !(sqrt(dx * dx + dy * dy + dz * dz) > range_limit);
            }
            if (ok) {
                int ix = lazygrid_index(px, cx, res, g.size[0]);
                int iy = lazygrid_index(py, cy, res, g.size[1]);
                int iz = lazygrid_index(pz, cz, res, g.size[2]);
                ok = ix >= 0 && ix < g.size[0] && iy >= 0 && iy < g.size[1] && iz >= 0 && iz < g.size[2];
                if (ok) {
                    slot = (ix * g.size[1] + iy) * g.size[2] + iz;
                    // offset from the cell origin in cell units, quantised to 2^-40
                    double ux = (px - (cx + (ix - g.size[0] / 2.0) * res)) / res;
                    double uy = (py - (cy + (iy - g.size[1] / 2.0) * res)) / res;
                    double uz = (pz - (cz + (iz - g.size[2] / 2.0) * res)) / res;
                    q[0] = __double2ll_rn(ux * S1);
                    q[1] = __double2ll_rn(uy * S1);
                    q[2] = __double2ll_rn(uz * S1);
                    double qx = (double)q[0] * IS1, qy = (double)q[1] * IS1, qz = (double)q[2] * IS1;
                    q[3] = __double2ll_rn(qx * qx * S2);
                    q[4] = __double2ll_rn(qx * qy * S2);
                    q[5] = __double2ll_rn(qx * qz * S2);
                    q[6] = __double2ll_rn(qy * qy * S2);
                    q[7] = __double2ll_rn(qy * qz * S2);
                    q[8] = __double2ll_rn(qz * qz * S2);
                }
            }
            if (!ok) dropped++;
        }
        // wavefront pre-reduction by cell: one round per distinct cell among the 64 lanes
        unsigned long long active = __ballot(slot >= 0);
        while (active) {
            int leader = __ffsll((long long)active) - 1;
            int s0 = __shfl(slot, leader, 64);
            bool mine = (slot == s0);
            unsigned long long mask = __ballot(mine);
            long long v[9];
#pragma unroll
            for (int k = 0; k < 9; k++) v[k] = wave_sum(mine ? q[k] : 0ll);
            if ((int)lane == leader) {
                int id = get_or_assign(table, s0, acc_slot, ctr, cap);
                if (id >= 0 && (uint32_t)id < cap) {
                    NdtAcc *a = acc + id;
                    __hip_atomic_fetch_add(&a->n, (unsigned long long)__popcll(mask), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                    for (int k = 0; k < 3; k++) atomic_add_ll(&a->s1[k], v[k]);
#pragma unroll
                    for (int k = 0; k < 6; k++) atomic_add_ll(&a->s2[k], v[3 + k]);
                }
            }
            active &= ~mask;
        }
    }
    if (dropped) atomicAdd(&s_dropped, dropped);
    __syncthreads();
    // atomics bypass the vector L1: drop lines that phase A cached before they were updated
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();

    // ---------------- phase B: moments -> Gaussian (in place in the scratch arena) -------------
    unsigned n_alloc = ctr->n_alloc;
    if (n_alloc > cap) n_alloc = cap;
    NdtCell *tmp = reinterpret_cast<NdtCell *>(acc);   // 80 B in, 80 B out
    for (unsigned id = tid; id < n_alloc; id += NDT_BUILD_THREADS) {
        NdtAcc a = acc[id];
        NdtCell c;
        c.n = 0;
        c.slot = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) c.mean[k] = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) c.cov[k] = 0;
        unsigned long long n = a.n;
        if (n >= 2 && n >= (unsigned long long)n_min) {
            unsigned slot = acc_slot[id];
            int iz = slot % g.size[2];
            int iy = (slot / g.size[2]) % g.size[1];
            int ix = slot / (g.size[2] * g.size[1]);
            double dn = (double)n;
            double m[3], S[6];
            for (int k = 0; k < 3; k++) m[k] = ((double)a.s1[k] / dn) * IS1;
            const double IS2 = 1.0 / S2;
            S[0] = (double)a.s2[0] * IS2 - dn * m[0] * m[0];
            S[1] = (double)a.s2[1] * IS2 - dn * m[0] * m[1];
            S[2] = (double)a.s2[2] * IS2 - dn * m[0] * m[2];
            S[3] = (double)a.s2[3] * IS2 - dn * m[1] * m[1];
            S[4] = (double)a.s2[4] * IS2 - dn * m[1] * m[2];
            S[5] = (double)a.s2[5] * IS2 - dn * m[2] * m[2];
            double sc = res * res / (dn - 1.0);
            double C[9] = {S[0] * sc, S[1] * sc, S[2] * sc, S[1] * sc, S[3] * sc, S[4] * sc, S[2] * sc, S[4] * sc, S[5] * sc};
            double ev[3], V[9];
            jacobi_eig<3>(3, C, ev, V);
            // NDTCell::rescaleCovariance
            if (ev[2] > 0 && ev[0] > NDT_DEGENERATE_REL * ev[2]) {
                bool recalc = false;
                double mx = ev[2];
                for (int k = 0; k < 3; k++)
                    if (mx > ev[k] * eval_factor) { ev[k] = mx / eval_factor; recalc = true; }
                if (recalc) {
                    for (int r = 0; r < 3; r++)
                        for (int q2 = r; q2 < 3; q2++) {
                            double s = 0;
                            for (int k = 0; k < 3; k++) s += V[r * 3 + k] * ev[k] * V[q2 * 3 + k];
                            C[r * 3 + q2] = s;
                        }
                }
                c.mean[0] = cx + (ix - g.size[0] / 2.0) * res + m[0] * res;
                c.mean[1] = cy + (iy - g.size[1] / 2.0) * res + m[1] * res;
                c.mean[2] = cz + (iz - g.size[2] / 2.0) * res + m[2] * res;
                c.cov[0] = C[0]; c.cov[1] = C[1]; c.cov[2] = C[2];
                c.cov[3] = C[4]; c.cov[4] = C[5]; c.cov[5] = C[8];
                c.n = (uint32_t)n;
                c.slot = slot;
            }
        }
        tmp[id] = c;
    }
    __syncthreads();

    // ---------------- phase C: rank Gaussian cells in slot order (block scan over the table) ---
    const unsigned wave = tid >> 6;
    for (unsigned sbase = 0; sbase < (unsigned)g.slots; sbase += NDT_BUILD_THREADS) {
        unsigned slot = sbase + tid;
        int id = (slot < (unsigned)g.slots) ? table[slot] : NDT_EMPTY;
        bool touched = id != NDT_EMPTY;
        bool valid = touched && (uint32_t)id < n_alloc && tmp[id].n > 0;
        unsigned long long bal = __ballot(valid);
        if (lane == 0) s_wave_cnt[wave] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before = s_base;
        for (unsigned w = 0; w < wave; w++) before += s_wave_cnt[w];
        unsigned total = 0;
        for (unsigned w = 0; w < NDT_BUILD_THREADS / 64; w++) total += s_wave_cnt[w];
        if (valid) {
            unsigned rank = before + (unsigned)__popcll(bal & lanemask_lt());
            cells[rank] = tmp[id];
            table[slot] = (int)rank;
        } else if (touched) {
            table[slot] = NDT_EMPTY;
        }
        __syncthreads();
        if (tid == 0) s_base += total;
    }
    __syncthreads();

    // ---------------- phase D: leave the scratch zeroed, publish counters ----------------------
    {
        unsigned long long *z = reinterpret_cast<unsigned long long *>(acc);
        for (unsigned k = tid; k < n_alloc * 10u; k += NDT_BUILD_THREADS) z[k] = 0ull;
    }
    if (tid == 0) {
        ctr->n_cells = s_base;
        ctr->n_alloc = 0;
        ctr->n_dropped = s_dropped;
    }
}

// Installs ready-made Gaussians (CellVector-like maps, KATs): cells must arrive sorted by slot,
// one per slot (the host wrapper guarantees it).
extern "C" __global__ void ndt_install_cells_kernel(NdtSetView set, unsigned map, const NdtCell *__restrict__ src,
                                                    unsigned n_cells)
{
    const NdtGrid g = set.grid;
    int32_t *table = set.table + (size_t)map * g.slots;
    NdtCell *cells = set.cells + (size_t)map * g.max_cells;
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_cells) {
        NdtCell c = src[i];
        cells[i] = c;
        table[c.slot] = (int)i;
    }
    if (i == 0) {
        set.counters[map].n_cells = n_cells;
        set.counters[map].n_alloc = 0;
        set.counters[map].overflow = 0;
        set.counters[map].n_dropped = 0;
    }
}

// reset the dense tables of the maps being rebuilt (-1 everywhere); must precede ndt_launch_build
hipError_t ndt_launch_table_reset(const NdtSetView &set, size_t first, size_t count, hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    return hipMemsetAsync(set.table + first * (size_t)set.grid.slots, 0xFF,
                          count * (size_t)set.grid.slots * sizeof(int32_t), stream);
}

hipError_t ndt_launch_build(const NdtSetView &set, size_t first, size_t count, const void *xyz_dev, size_t n_points,
                            size_t stride_bytes, size_t map_stride_bytes, double range_limit,
                            const double *range_origins_dev, int n_min, double eval_factor, hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    // second-moment scale: N * max|u_a u_b| * 2^shift < 2^63 with |u| < 2  ->  shift <= 61 - ceil(log2 N)
    int lg = 1;
    while ((1ull << lg) < (unsigned long long)(n_points ? n_points : 1)) lg++;
    int s2_shift = 61 - lg;
    if (s2_shift > 46) s2_shift = 46;
    if (s2_shift < 20) s2_shift = 20;
    hipLaunchKernelGGL(ndt_build_kernel, dim3((unsigned)count), dim3(NDT_BUILD_THREADS), 0, stream, set, (unsigned)first,
                       (const char *)xyz_dev, (unsigned)n_points, (unsigned)stride_bytes, map_stride_bytes, range_limit,
                       range_origins_dev, n_min, eval_factor, s2_shift);
    return hipGetLastError();
}

hipError_t ndt_launch_install_cells(const NdtSetView &set, size_t map, const double *mean3_dev, const double *cov9_dev,
                                    size_t n_cells, hipStream_t stream)
{
    (void)mean3_dev; (void)cov9_dev;   // the host wrapper passes packed NdtCell records through mean3_dev
    hipError_t e = hipMemsetAsync(set.table + map * (size_t)set.grid.slots, 0xFF,
                                  (size_t)set.grid.slots * sizeof(int32_t), stream);
    if (e != hipSuccess) return e;
    unsigned blocks = (unsigned)((n_cells + 255) / 256);
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(ndt_install_cells_kernel, dim3(blocks), dim3(256), 0, stream, set, (unsigned)map,
                       reinterpret_cast<const NdtCell *>(mean3_dev), (unsigned)n_cells);
    return hipGetLastError();
}
