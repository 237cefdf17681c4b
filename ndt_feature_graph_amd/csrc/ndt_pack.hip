// ndt_pack.hip -- device-side exchange records of cell maps (product code).
//
// SURVEY.md 8(e) phases A-B: with the node maps of a graph replay built data-parallel (node k on rank k mod world), every
// rank needs every node map before it registers its share of the edges (ndt_feature_graph.cpp:273 reads nodes_[ref] and
// nodes_[mov]).  What travels is what the matcher reads -- the Gaussian cells in slot order -- plus, when the overlap score
// of ndt_feature_node.h:213-252 is wanted, the occupancy of every cell.  One fixed-stride record per map so that ONE
// all_gather moves all of them:
//     ndtgpu_packed_header {n_cells, flags, n_dropped, cells_cap}            16 bytes
//     cells_cap x ndtgpu_cell_record (= NdtCell, 80 bytes; the first n_cells are valid, in slot order)
//     slots x float occupancy                                                  (with_occupancy only)
// Unpacking installs the cells, rebuilds the rank map (the only index the matcher probes) and the counters: a map that
// was unpacked is indistinguishable from the map that was packed.
#include "ndt_math.h"

#define NDT_PACK_THREADS 256
#define NDT_PACK_F_OVERFLOW 1u      // the map overflowed max_cells where it was built, or holds more cells than cells_cap
#define NDT_PACK_F_OCC 2u           // the record carries occupancies

struct NdtPackedHeader { uint32_t n_cells, flags, n_dropped, cells_cap; };
static_assert(sizeof(NdtPackedHeader) == 16, "packed header");

// one workgroup per map
extern "C" __global__ __launch_bounds__(NDT_PACK_THREADS) void ndt_pack_kernel(NdtSetView set, unsigned first,
                                                                              char *__restrict__ buf, size_t stride,
                                                                              unsigned cells_cap, int with_occ)
{
    const unsigned map = first + blockIdx.x, tid = threadIdx.x;
    const NdtGrid g = set.grid;
    char *rec = buf + (size_t)blockIdx.x * stride;
    const NdtMapCounters c = set.counters[map];
    const unsigned n = c.n_cells > g.max_cells ? g.max_cells : c.n_cells;
    const unsigned n_out = n > cells_cap ? cells_cap : n;
    if (tid == 0) {
        NdtPackedHeader h;
        h.n_cells = n_out;
        h.flags = ((c.overflow || n > cells_cap) ? NDT_PACK_F_OVERFLOW : 0u) | (with_occ ? NDT_PACK_F_OCC : 0u);
        h.n_dropped = c.n_dropped;
        h.cells_cap = cells_cap;
        *reinterpret_cast<NdtPackedHeader *>(rec) = h;
    }
    // 80-byte records as 16-byte pieces: coalesced on both sides
    const uint4 *src = reinterpret_cast<const uint4 *>(ndt_cells_of(set, map, set.cell_sel ? set.cell_sel[map] : 0u));
    uint4 *dst = reinterpret_cast<uint4 *>(rec + sizeof(NdtPackedHeader));
    for (unsigned i = tid; i < n_out * 5u; i += NDT_PACK_THREADS) dst[i] = src[i];
    if (with_occ) {
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
    const bool occ_fits = sizeof(NdtPackedHeader) + (size_t)h.cells_cap * sizeof(NdtCell) + (size_t)g.slots * sizeof(float) <= stride;
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
        const float *o = reinterpret_cast<const float *>(rec + sizeof(NdtPackedHeader) + (size_t)h.cells_cap * sizeof(NdtCell));
        float *od = set.occ + (size_t)map * g.slots;
        for (unsigned i = tid; i < (unsigned)g.slots; i += NDT_PACK_THREADS) od[i] = o[i];
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

hipError_t ndt_launch_pack(const NdtSetView &set, size_t first, size_t count, void *buf_dev, size_t stride, unsigned cells_cap,
                           int with_occ, hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(ndt_pack_kernel, dim3((unsigned)count), dim3(NDT_PACK_THREADS), 0, stream, set, (unsigned)first,
                       (char *)buf_dev, stride, cells_cap, with_occ);
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
