// ndt_pack.hip -- device-side exchange records of cell maps (product code).
//
// SURVEY.md 8(e) phases A-B: with the node maps of a graph replay built data-parallel (node k on rank k mod world), every
// rank needs every node map before it registers its share of the edges (ndt_feature_graph.cpp:273 reads nodes_[ref] and
// nodes_[mov]).  What travels is what the matcher reads -- the Gaussian cells in slot order -- plus, when the overlap score
// of ndt_feature_node.h:213-252 is wanted, the occupancy of every cell.  One fixed-stride record per map so that ONE
// all_gather moves all of them:
//     ndtgpu_packed_header {n_cells, flags, n_dropped, cells_cap}            16 bytes
//     cells_cap x ndtgpu_cell_record (= NdtCell, 80 bytes; the first n_cells are valid, in slot order)
//     slots x float occupancy                                                  (with_occupancy, dense form)
//  or {n_occ, occ_cap} + occ_cap x {slot, occupancy}                            (sparse form: the cells that have a reading,
//                                                                                in slot order -- a fused node map of the
//                                                                                replay has readings in 2-3 % of its 80 k
//                                                                                slots: 22 KB instead of 320 KB per node)
// Unpacking installs the cells, rebuilds the rank map (the only index the matcher probes) and the counters: a map that
// was unpacked is indistinguishable from the map that was packed.
#include "ndt_math.h"
#include "ndt_wave.h"

#define NDT_PACK_THREADS 256
#define NDT_PACK_F_OVERFLOW 1u      // the map overflowed max_cells where it was built, or holds more cells than cells_cap
#define NDT_PACK_F_OCC 2u           // the record carries occupancies

#define NDT_PACK_F_OCC_SPARSE 4u    // ... as (slot, value) pairs of the cells that have a reading

struct NdtPackedHeader { uint32_t n_cells, flags, n_dropped, cells_cap; };
static_assert(sizeof(NdtPackedHeader) == 16, "packed header");
struct NdtPackedOccHead { uint32_t n_occ, occ_cap; };
struct NdtPackedOcc { uint32_t slot; float occ; };
static_assert(sizeof(NdtPackedOccHead) == 8 && sizeof(NdtPackedOcc) == 8, "sparse occupancy block");

// Cells of a map that have a reading (occupancy != 0), counted by the 4 waves of a workgroup: wave w takes the w-th quarter
// of the slots, 64 consecutive slots per step (coalesced), one ballot per step.  Returns the wave's count (wave-uniform).
static __device__ __forceinline__ unsigned occ_count_quarter(const float *o, unsigned begin, unsigned end)
{
    const unsigned lane = threadIdx.x & 63u;
    unsigned n = 0;
    for (unsigned b = begin; b < end; b += 64u) {
        const unsigned s = b + lane;
        const bool on = s < end && o[s] != 0.0f;
        n += (unsigned)__popcll(ndt_ballot(on));
    }
    return n;
}

// one workgroup per map: counts[blockIdx.x] = its cells with a reading (maps: first + blockIdx.x, or maps[blockIdx.x])
extern "C" __global__ __launch_bounds__(NDT_PACK_THREADS) void ndt_occ_count_kernel(NdtSetView set, unsigned first,
                                                                                   const uint32_t *__restrict__ maps,
                                                                                   unsigned *__restrict__ counts)
{
    __shared__ unsigned s_n[NDT_PACK_THREADS / 64];
    const unsigned map = maps ? maps[blockIdx.x] : first + blockIdx.x, wave = threadIdx.x >> 6;
    const unsigned slots = (unsigned)set.grid.slots, q = ((slots + 3u) / 4u + 63u) & ~63u;
    const float *o = set.occ + (size_t)map * slots;
    const unsigned n = occ_count_quarter(o, min(slots, wave * q), min(slots, (wave + 1u) * q));
    if ((threadIdx.x & 63u) == 0u) s_n[wave] = n;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_n[0] + s_n[1] + s_n[2] + s_n[3];
}

// The (slot, occupancy) pairs of the cells with a reading of map `map`, in slot order, to pairs[0 .. min(total, cap)); returns
// the total (workgroup-uniform).  All NDT_PACK_THREADS threads; s_n: one word per wave.
static __device__ __forceinline__ unsigned occ_compact(const NdtSetView &set, unsigned map, NdtPackedOcc *pairs, unsigned cap, unsigned *s_n)
{
    // count per wave, then every wave writes its quarter's pairs behind those of the waves before it
    const unsigned tid = threadIdx.x, slots = (unsigned)set.grid.slots, q = ((slots + 3u) / 4u + 63u) & ~63u, wave = tid >> 6, lane = tid & 63u;
    const float *o = set.occ + (size_t)map * slots;
    const unsigned begin = min(slots, wave * q), end = min(slots, (wave + 1u) * q);
    const unsigned mine = occ_count_quarter(o, begin, end);
    if (lane == 0u) s_n[wave] = mine;
    __syncthreads();
    unsigned at = 0, total = 0;
    for (unsigned w = 0; w < NDT_PACK_THREADS / 64; w++) { if (w < wave) at += s_n[w]; total += s_n[w]; }
    for (unsigned b = begin; b < end; b += 64u) {
        const unsigned sl = b + lane;
        const float v = sl < end ? o[sl] : 0.0f;
        const unsigned long long m = ndt_ballot(v != 0.0f);
        const unsigned pos = at + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        if (v != 0.0f && pos < cap) pairs[pos] = NdtPackedOcc{sl, v};
        at += (unsigned)__popcll(m);
    }
    return total;
}

// one workgroup per map maps[blockIdx.x]: its pairs to pairs + offs[blockIdx.x] (room for offs[blockIdx.x + 1] - offs[blockIdx.x])
extern "C" __global__ __launch_bounds__(NDT_PACK_THREADS) void ndt_occ_list_kernel(NdtSetView set, const uint32_t *__restrict__ maps,
                                                                                  const unsigned *__restrict__ offs,
                                                                                  uint2 *__restrict__ pairs)
{
    __shared__ unsigned s_n[NDT_PACK_THREADS / 64];
    const unsigned o0 = offs[blockIdx.x], o1 = offs[blockIdx.x + 1u];
    (void)occ_compact(set, maps[blockIdx.x], reinterpret_cast<NdtPackedOcc *>(pairs) + o0, o1 - o0, s_n);
}

// one workgroup per map
extern "C" __global__ __launch_bounds__(NDT_PACK_THREADS) void ndt_pack_kernel(NdtSetView set, unsigned first,
                                                                              char *__restrict__ buf, size_t stride,
                                                                              unsigned cells_cap, int with_occ, unsigned occ_cap)
{
    __shared__ unsigned s_n[NDT_PACK_THREADS / 64];
    const unsigned map = first + blockIdx.x, tid = threadIdx.x;
    const NdtGrid g = set.grid;
    char *rec = buf + (size_t)blockIdx.x * stride;
    const NdtMapCounters c = set.counters[map];
    const unsigned n = c.n_cells > g.max_cells ? g.max_cells : c.n_cells;
    const unsigned n_out = n > cells_cap ? cells_cap : n;
    bool occ_cut = false;
    if (with_occ == 2) {
        // sparse occupancies, in slot order: the record's bytes do not depend on timing
        char *ob = rec + sizeof(NdtPackedHeader) + (size_t)cells_cap * sizeof(NdtCell);
        const unsigned total = occ_compact(set, map, reinterpret_cast<NdtPackedOcc *>(ob + sizeof(NdtPackedOccHead)), occ_cap, s_n);
        occ_cut = total > occ_cap;
        if (tid == 0) *reinterpret_cast<NdtPackedOccHead *>(ob) = NdtPackedOccHead{occ_cut ? occ_cap : total, occ_cap};
    }
    if (tid == 0) {
        NdtPackedHeader h;
        h.n_cells = n_out;
        h.flags = ((c.overflow || n > cells_cap || occ_cut) ? NDT_PACK_F_OVERFLOW : 0u) | (with_occ ? NDT_PACK_F_OCC : 0u) |
                  (with_occ == 2 ? NDT_PACK_F_OCC_SPARSE : 0u);
        h.n_dropped = c.n_dropped;
        h.cells_cap = cells_cap;
        *reinterpret_cast<NdtPackedHeader *>(rec) = h;
    }
    // 80-byte records as 16-byte pieces: coalesced on both sides
    const uint4 *src = reinterpret_cast<const uint4 *>(ndt_cells_of(set, map, set.cell_sel ? set.cell_sel[map] : 0u));
    uint4 *dst = reinterpret_cast<uint4 *>(rec + sizeof(NdtPackedHeader));
    for (unsigned i = tid; i < n_out * 5u; i += NDT_PACK_THREADS) dst[i] = src[i];
    if (with_occ == 1) {
        const float *o = set.occ + (size_t)map * g.slots;
        float *od = reinterpret_cast<float *>(rec + sizeof(NdtPackedHeader) + (size_t)cells_cap * sizeof(NdtCell));
        for (unsigned i = tid; i < (unsigned)g.slots; i += NDT_PACK_THREADS) od[i] = o[i];
    }
}

extern "C" __global__ __launch_bounds__(NDT_PACK_THREADS) void ndt_unpack_kernel(NdtSetView set, unsigned first,
                                                                                const char *__restrict__ buf, size_t stride,
                                                                                int with_occ)
{
    const unsigned map = first + blockIdx.x, tid = threadIdx.x;
    const NdtGrid g = set.grid;
    const char *rec = buf + (size_t)blockIdx.x * stride;
    const NdtPackedHeader h = *reinterpret_cast<const NdtPackedHeader *>(rec);
    unsigned n = h.n_cells;
    // the header is data from elsewhere: never read past the record -- n_cells <= cells_cap, both inside `stride`
    const size_t room = stride > sizeof(NdtPackedHeader) ? (stride - sizeof(NdtPackedHeader)) / sizeof(NdtCell) : 0;
    const unsigned cap_rec = (unsigned)(h.cells_cap < room ? h.cells_cap : room);
    const bool too_many = n > g.max_cells || n > cap_rec;
    if (n > cap_rec) n = cap_rec;
    if (n > g.max_cells) n = g.max_cells;
    const size_t occ_at = sizeof(NdtPackedHeader) + (size_t)h.cells_cap * sizeof(NdtCell);
    const bool sparse = (h.flags & NDT_PACK_F_OCC_SPARSE) != 0u;
    NdtPackedOccHead oh = {0u, 0u};
    if (sparse && (h.flags & NDT_PACK_F_OCC) && occ_at + sizeof(NdtPackedOccHead) <= stride)
        oh = *reinterpret_cast<const NdtPackedOccHead *>(rec + occ_at);
    const bool occ_fits = sparse ? (occ_at + sizeof(NdtPackedOccHead) + (size_t)oh.occ_cap * sizeof(NdtPackedOcc) <= stride && oh.n_occ <= oh.occ_cap)
                                 : (occ_at + (size_t)g.slots * sizeof(float) <= stride);
    uint2 *rankmap = set.rankmap + (size_t)map * ndt_rm_stride(g);
    const unsigned bm_words = (unsigned)((g.slots + 31) >> 5);
    NdtCell *cells = set.cells + (size_t)map * g.max_cells;          // an installed map lives in the first cell array
    const NdtCell *src = reinterpret_cast<const NdtCell *>(rec + sizeof(NdtPackedHeader));
    // 1. the rank map forgets the map that was here
    for (unsigned w = tid; w < bm_words; w += NDT_PACK_THREADS) rankmap[w] = make_uint2(0u, 0u);
    {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *d4 = reinterpret_cast<uint4 *>(cells);
        for (unsigned i = tid; i < n * 5u; i += NDT_PACK_THREADS) d4[i] = s4[i];
    }
    __syncthreads();                                                   // (the zeroes have reached the L2: write-through)
    // 2. a bit per cell, the rank of the first cell of every 32-slot word (the cells arrive in slot order)
    for (unsigned i = tid; i < n; i += NDT_PACK_THREADS) {
        const unsigned slot = src[i].slot;
        if (slot >= (unsigned)g.slots) continue;                       // (a record of another grid geometry: dropped)
        atomicOr(&rankmap[slot >> 5].x, 1u << (slot & 31u));
        if (i == 0 || (src[i - 1].slot >> 5) != (slot >> 5)) rankmap[slot >> 5].y = i;
    }
    if (with_occ && set.occ && (h.flags & NDT_PACK_F_OCC) && occ_fits) {
        float *od = set.occ + (size_t)map * g.slots;
        if (sparse) {
            // every cell without a reading, then the pairs (another thread's store to the same word: behind the barrier)
            for (unsigned i = tid; i < (unsigned)g.slots; i += NDT_PACK_THREADS) od[i] = 0.0f;
            __syncthreads();
            const NdtPackedOcc *pairs = reinterpret_cast<const NdtPackedOcc *>(rec + occ_at + sizeof(NdtPackedOccHead));
            for (unsigned i = tid; i < oh.n_occ; i += NDT_PACK_THREADS) {
                const NdtPackedOcc pr = pairs[i];
                if (pr.slot < (unsigned)g.slots) od[pr.slot] = pr.occ;
            }
        } else {
            const float *o = reinterpret_cast<const float *>(rec + occ_at);
            for (unsigned i = tid; i < (unsigned)g.slots; i += NDT_PACK_THREADS) od[i] = o[i];
        }
    }
    if (tid == 0) {
        NdtMapCounters c = set.counters[map];
        c.n_cells = n;
        c.n_alloc = 0;
        // (a record that had to be clamped, or whose occupancy block does not fit its stride, is flagged like an overflow:
        //  the matcher refuses the map)
        c.overflow = ((h.flags & NDT_PACK_F_OVERFLOW) || too_many || (with_occ && (h.flags & NDT_PACK_F_OCC) && !occ_fits)) ? 1u : 0u;
        c.n_dropped = h.n_dropped;
        set.counters[map] = c;
        if (set.cell_sel) set.cell_sel[map] = 0u;
    }
}

// with_occ: 0 no occupancies, 1 one float per slot, 2 (slot, value) pairs of the cells with a reading, at most occ_cap
hipError_t ndt_launch_pack(const NdtSetView &set, size_t first, size_t count, void *buf_dev, size_t stride, unsigned cells_cap,
                           int with_occ, unsigned occ_cap, hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(ndt_pack_kernel, dim3((unsigned)count), dim3(NDT_PACK_THREADS), 0, stream, set, (unsigned)first,
                       (char *)buf_dev, stride, cells_cap, with_occ, occ_cap);
    return hipGetLastError();
}

size_t ndt_pack_sparse_occ_bytes(unsigned occ_cap) { return sizeof(NdtPackedOccHead) + (size_t)occ_cap * sizeof(NdtPackedOcc); }

// maps_dev: the maps to count (count of them), or NULL for maps [first, first + count)
hipError_t ndt_launch_occ_count(const NdtSetView &set, size_t first, const uint32_t *maps_dev, size_t count, unsigned *counts_dev,
                                hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(ndt_occ_count_kernel, dim3((unsigned)count), dim3(NDT_PACK_THREADS), 0, stream, set, (unsigned)first, maps_dev,
                       counts_dev);
    return hipGetLastError();
}

// the (slot, occupancy) pairs of maps_dev[u] to pairs_dev + offs_dev[u], u < count (offs_dev: count + 1 offsets)
hipError_t ndt_launch_occ_list(const NdtSetView &set, const uint32_t *maps_dev, size_t count, const unsigned *offs_dev, void *pairs_dev,
                               hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(ndt_occ_list_kernel, dim3((unsigned)count), dim3(NDT_PACK_THREADS), 0, stream, set, maps_dev, offs_dev,
                       (uint2 *)pairs_dev);
    return hipGetLastError();
}

hipError_t ndt_launch_unpack(const NdtSetView &set, size_t first, size_t count, const void *buf_dev, size_t stride, int with_occ,
                             hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(ndt_unpack_kernel, dim3((unsigned)count), dim3(NDT_PACK_THREADS), 0, stream, set, (unsigned)first,
                       (const char *)buf_dev, stride, with_occ);
    return hipGetLastError();
}
