// ndtgpu_api.hip -- C-ABI (include/ndtgpu.h) over the HIP kernels.  Host side only: handle and
// arena management, argument checking, staging copies.  No CPU compute path exists here: without a
// device every compute entry point fails with NDTGPU_ERR_NO_DEVICE.
#include "../../include/ndtgpu.h"
#include "ndt_math.h"
#include "ndt_solver.h"
#include "ndt_pose.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <atomic>
#include <functional>
#include <mutex>
#include <chrono>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;

ndtgpu_status fail(ndtgpu_status s, const char *what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return s;
}

#define HIP_TRY(expr)                                                       \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) return fail(NDTGPU_ERR_HIP, #expr, _e);       \
    } while (0)

bool have_device()
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

}  // namespace

struct ndtgpu_mapset {
    NdtSetView v{};
    size_t n_maps = 0;
    std::vector<double> centres_host;
    std::vector<unsigned char> nice_host;   // per map: fp32 cell offsets are exact (ndt_grid_is_nice)
    int nice_range(size_t first, size_t count) const
    {
        for (size_t m = first; m < first + count; m++)
            if (!nice_host[m]) return 0;
        return 1;
    }
    // Streams that may still hold work on this set (writers: builds, unpack, add_cloud; readers: matcher launches): one event
    // per recently used stream, recorded AFTER the launch.  The host-synchronous entries wait for these events -- not for
    // stream handles, which the caller may have destroyed since, and not only for the last writer (a matcher that still reads
    // the maps on another stream is waited for as well).
    struct StreamMark { hipStream_t st; hipEvent_t ev; };
    std::vector<StreamMark> marks;
    bool null_stream_used = false;       // the null stream needs no event: its handle is always valid (and an event record
                                         // costs the reference's one-pair-at-a-time call shape ~10 us of its 0.37 ms)
    ndtgpu_status touch(hipStream_t st)
    {
        if (st == nullptr) { null_stream_used = true; return NDTGPU_OK; }
        for (StreamMark &m : marks)
            if (m.st == st) { HIP_TRY(hipEventRecord(m.ev, st)); return NDTGPU_OK; }
        if (marks.size() >= 8) {                 // many streams over time: retire the oldest entry once its work is done
            HIP_TRY(hipEventSynchronize(marks.front().ev));
            (void)hipEventDestroy(marks.front().ev);
            marks.erase(marks.begin());
        }
        StreamMark m{st, nullptr};
        HIP_TRY(hipEventCreateWithFlags(&m.ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(m.ev, st));
        marks.push_back(m);
        return NDTGPU_OK;
    }
    // ... for work that is about to be enqueued on `st`: what was recorded on `st` itself is ordered by the stream
    ndtgpu_status wait_all_on(hipStream_t st)
    {
        if (null_stream_used && st != nullptr) { HIP_TRY(hipStreamSynchronize(nullptr)); null_stream_used = false; }
        for (StreamMark &m : marks)
            if (m.st != st) HIP_TRY(hipEventSynchronize(m.ev));
        return NDTGPU_OK;
    }
    ndtgpu_status wait_all()
    {
        if (null_stream_used) { HIP_TRY(hipStreamSynchronize(nullptr)); null_stream_used = false; }
        for (StreamMark &m : marks) HIP_TRY(hipEventSynchronize(m.ev));
        return NDTGPU_OK;
    }
    // staging buffers reused across calls
    void *stage = nullptr;
    size_t stage_bytes = 0;
    double *origins_dev = nullptr;
    size_t origins_cap = 0;
    hipEvent_t origins_ev = nullptr;   // recorded after the last launch that reads origins_dev (it may be on another stream)
    bool origins_ev_valid = false;
    // room for `n` doubles in origins_dev, ordered behind its last reader: `st` waits for that launch before the buffer is
    // overwritten (or the host does, before it is replaced)
    ndtgpu_status origins_reserve(size_t n, hipStream_t st)
    {
        if (origins_cap < n) {
            if (origins_ev_valid) HIP_TRY(hipEventSynchronize(origins_ev));
            if (origins_dev) (void)hipFree(origins_dev);
            origins_dev = nullptr;
            origins_cap = 0;
            HIP_TRY(hipMalloc((void **)&origins_dev, n * sizeof(double)));
            origins_cap = n;
        } else if (origins_ev_valid) {
            HIP_TRY(hipStreamWaitEvent(st, origins_ev, 0));
        }
        return NDTGPU_OK;
    }
    ndtgpu_status origins_used(hipStream_t st)
    {
        if (!origins_ev) HIP_TRY(hipEventCreateWithFlags(&origins_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(origins_ev, st));
        origins_ev_valid = true;
        return NDTGPU_OK;
    }
    // workgroups the persistent matcher launches with this set as target get at most (0: one per CU).  The registrar keeps its
    // matcher launches on part of the chip: the rest stays free for the next sub-batch's builds while a launch runs
    unsigned match_groups = 0;
    // matcher work area: ticket counters, parked list, parked solver states
    void *work = nullptr;
    size_t work_bytes = 0;
    hipEvent_t work_ev = nullptr;      // recorded after the last launch that uses `work`
    bool work_ev_valid = false;
    hipStream_t work_stream = nullptr;
    // profiling hooks: [0,1] bracket the build kernel, [2,3] the match kernel
    bool profiling = false;
    bool profile_span = false;         // a chunked host build is ONE bracket: the chunks' launches do not re-record the events
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ev_valid[2] = {false, false};

    ndtgpu_status ensure_work(size_t bytes)
    {
        if (bytes <= work_bytes) return NDTGPU_OK;
        if (work) (void)hipFree(work);
        work = nullptr;
        work_bytes = 0;
        HIP_TRY(hipMalloc(&work, bytes));
        work_bytes = bytes;
        return NDTGPU_OK;
    }
    // work area of the grid-barrier matcher (a control block + partial sums per registration); its kernels leave the
    // control blocks zeroed, so a call only clears what it cannot know to be clean
    void *coop_work = nullptr;
    size_t coop_bytes = 0, coop_clean_stride = 0, coop_clean_upto = 0;
    ndtgpu_status ensure_coop(size_t bytes);     // (defined below: waits for the last grid-barrier launch before it frees)
    ndtgpu_status ensure_coop_impl(size_t bytes)
    {
        if (bytes <= coop_bytes) return NDTGPU_OK;
        if (coop_work) (void)hipFree(coop_work);
        coop_work = nullptr;
        coop_bytes = 0;
        coop_clean_upto = 0;
        HIP_TRY(hipMalloc(&coop_work, bytes));
        coop_bytes = bytes;
        return NDTGPU_OK;
    }
    // pinned host mirror of small staging blocks (poses, indices, results of a host-pointer matcher call): copies from /
    // to pinned memory are truly asynchronous and skip the runtime's own bounce buffer
    void *pin = nullptr;
    size_t pin_bytes = 0;
    ndtgpu_status ensure_pin(size_t bytes)
    {
        if (bytes <= pin_bytes) return NDTGPU_OK;
        if (pin) (void)hipHostFree(pin);
        pin = nullptr;
        pin_bytes = 0;
        HIP_TRY(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
        pin_bytes = bytes;
        return NDTGPU_OK;
    }
    // Host clouds (the reference's call sites hand over pcl::PointCloud on the host): a ring of pinned slots that host
    // threads fill from the caller's pageable memory while earlier slots travel to the device and earlier chunks of
    // maps are being built (stage_host_clouds below).  The copies run on a stream of their own.
    static constexpr int HOST_SLOTS = 6;
    static constexpr size_t HOST_SLOT_BYTES = 16u << 20;
    void *host_ring[HOST_SLOTS] = {};
    hipEvent_t host_ev[HOST_SLOTS] = {};
    bool host_ev_used[HOST_SLOTS] = {};
    hipStream_t host_copy_stream = nullptr;
    hipStream_t host_build_stream = nullptr;  // the synchronous host-cloud entries build on a stream of their own (no device-wide wait)
    ndtgpu_status ensure_host_build_stream()
    {
        if (!host_build_stream) HIP_TRY(hipStreamCreateWithFlags(&host_build_stream, hipStreamNonBlocking));
        return NDTGPU_OK;
    }
    hipEvent_t stage_free_ev = nullptr;      // recorded after the last kernel that reads the staged clouds
    bool stage_free_valid = false;
    ndtgpu_status ensure_host_ring()
    {
        if (host_copy_stream) return NDTGPU_OK;
        for (int k = 0; k < HOST_SLOTS; k++) {
            HIP_TRY(hipHostMalloc(&host_ring[k], HOST_SLOT_BYTES, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&host_ev[k], hipEventDisableTiming));
        }
        HIP_TRY(hipEventCreateWithFlags(&stage_free_ev, hipEventDisableTiming));
        HIP_TRY(hipStreamCreateWithFlags(&host_copy_stream, hipStreamNonBlocking));
        return NDTGPU_OK;
    }
    ndtgpu_status ensure_stage(size_t bytes)
    {
        if (bytes <= stage_bytes) return NDTGPU_OK;
        if (stage_free_valid) { HIP_TRY(hipEventSynchronize(stage_free_ev)); stage_free_valid = false; }
        if (stage) (void)hipFree(stage);
        stage = nullptr;
        stage_bytes = 0;
        HIP_TRY(hipMalloc(&stage, bytes));
        stage_bytes = bytes;
        return NDTGPU_OK;
    }
};

extern "C" {

// (bumped whenever a kernel changes: bench.py only quotes PMC figures taken with the same version)
const char *ndtgpu_version(void) { return "ndtgpu 0.6.6 (gfx950)"; }
const char *ndtgpu_last_error(void) { return g_err.c_str(); }

int ndtgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void ndtgpu_default_cell_params(ndtgpu_cell_params *p)
{
    p->n_min = 3;
    p->eval_factor = 1000.0;
}

void ndtgpu_default_match_params(ndtgpu_match_params *p)
{
    // the fuser preset: ndt_feature_fuser_hmt.cpp:356-357 with the production launch values
    p->n_neighbours = 2;
    p->itr_max = 30;
    p->delta_score = 1e-6;
    p->step_control = 1;
    p->lfd1 = 1.0;
    p->lfd2 = 0.05;
    p->dof_mask = 0x3f;
    p->use_initial_guess = 1;
}

const char *ndtgpu_kernel_name(int which)
{
    switch (which) {
    case 0: return "ndt_build_kernel";
    case 1: return "ndt_match_kernel";
    case 2: return "ndt_derivatives_kernel";
    default: return "";
    }
}

ndtgpu_status ndtgpu_mapset_create(const ndtgpu_grid_params *grid, size_t n_maps, ndtgpu_mapset **out)
{
    if (!grid || !out || n_maps == 0 || !(grid->res > 0)) return fail(NDTGPU_ERR_INVALID, "mapset_create: bad argument");
    if (!have_device()) return fail(NDTGPU_ERR_NO_DEVICE, "mapset_create: no HIP device");
    ndtgpu_mapset *s = new (std::nothrow) ndtgpu_mapset();
    if (!s) return fail(NDTGPU_ERR_ALLOC, "mapset_create: host alloc");
    s->n_maps = n_maps;
    s->v.n_maps = (uint32_t)n_maps;
    NdtGrid &g = s->v.grid;
    g.res = grid->res;
    long long slots = 1;
    for (int a = 0; a < 3; a++) {
        // LazyGrid::initialize: sizeX = abs(ceil(sizeXmeters / cellSizeX))
        g.size[a] = std::abs((int)std::ceil(grid->size[a] / grid->res));
        if (g.size[a] <= 0) { delete s; return fail(NDTGPU_ERR_INVALID, "mapset_create: empty grid axis"); }
        slots *= g.size[a];
    }
    if (slots > (1ll << 30)) { delete s; return fail(NDTGPU_ERR_INVALID, "mapset_create: grid too large"); }
    g.slots = (int)slots;
    for (int a = 0; a < 3; a++) g.half[a] = g.size[a] / 2.0;
    uint32_t cap = grid->max_cells ? grid->max_cells : (uint32_t)std::min<long long>(slots, 16384);
    if (cap > (1u << 24) - 1) cap = (1u << 24) - 1;
    g.max_cells = cap;
    s->centres_host.resize(n_maps * 3);
    for (size_t m = 0; m < n_maps; m++)
        for (int a = 0; a < 3; a++) s->centres_host[m * 3 + a] = grid->centre[a];
    s->nice_host.assign(n_maps, ndt_grid_is_nice(g, grid->centre) ? 1 : 0);

    hipError_t e;
#define ALLOC(ptr, bytes)                                                          \
    if ((e = hipMalloc((void **)&(ptr), (bytes))) != hipSuccess) {                 \
        ndtgpu_mapset_destroy(s);                                                  \
        return fail(NDTGPU_ERR_ALLOC, "mapset_create: hipMalloc " #ptr, e);        \
    }
    ALLOC(s->v.rankmap, n_maps * ndt_rm_stride(g) * sizeof(uint2));
    ALLOC(s->v.wtable, n_maps * (size_t)g.slots * sizeof(int32_t));
    ALLOC(s->v.bitmap, n_maps * (size_t)((g.slots + 31) / 32) * sizeof(uint32_t));
    ALLOC(s->v.cells, n_maps * (size_t)cap * sizeof(NdtCell));
    ALLOC(s->v.acc, n_maps * (size_t)cap * sizeof(NdtAcc));
    ALLOC(s->v.acc_slot, n_maps * (size_t)cap * sizeof(uint32_t));
    ALLOC(s->v.rank_agg, n_maps * (size_t)(NDT_RANK_SEGS + 2) * sizeof(uint32_t));
    ALLOC(s->v.counters, n_maps * sizeof(NdtMapCounters));
    ALLOC(s->v.centres, n_maps * 3 * sizeof(double));
#undef ALLOC
    if ((e = hipMemset(s->v.rankmap, 0, n_maps * ndt_rm_stride(g) * sizeof(uint2))) != hipSuccess ||
        (e = hipMemset(s->v.wtable, 0xFF, n_maps * (size_t)g.slots * sizeof(int32_t))) != hipSuccess ||
        (e = hipMemset(s->v.bitmap, 0, n_maps * (size_t)((g.slots + 31) / 32) * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMemset(s->v.acc, 0, n_maps * (size_t)cap * sizeof(NdtAcc))) != hipSuccess ||
        (e = hipMemset(s->v.counters, 0, n_maps * sizeof(NdtMapCounters))) != hipSuccess ||
        (e = hipMemset(s->v.rank_agg, 0, n_maps * (size_t)(NDT_RANK_SEGS + 2) * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMemcpy(s->v.centres, s->centres_host.data(), n_maps * 3 * sizeof(double), hipMemcpyHostToDevice)) !=
            hipSuccess) {
        ndtgpu_mapset_destroy(s);
        return fail(NDTGPU_ERR_HIP, "mapset_create: init", e);
    }
    *out = s;
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_mapset_destroy(ndtgpu_mapset *s)
{
    if (!s) return NDTGPU_OK;
    (void)hipDeviceSynchronize();
    if (s->pin) (void)hipHostFree(s->pin);
    for (auto &m : s->marks) (void)hipEventDestroy(m.ev);
    for (int k = 0; k < ndtgpu_mapset::HOST_SLOTS; k++) {
        if (s->host_ring[k]) (void)hipHostFree(s->host_ring[k]);
        if (s->host_ev[k]) (void)hipEventDestroy(s->host_ev[k]);
    }
    if (s->stage_free_ev) (void)hipEventDestroy(s->stage_free_ev);
    if (s->host_copy_stream) (void)hipStreamDestroy(s->host_copy_stream);
    if (s->host_build_stream) (void)hipStreamDestroy(s->host_build_stream);
    if (s->v.rankmap) (void)hipFree(s->v.rankmap);
    if (s->v.wtable) (void)hipFree(s->v.wtable);
    if (s->v.bitmap) (void)hipFree(s->v.bitmap);
    if (s->v.cells) (void)hipFree(s->v.cells);
    if (s->v.acc) (void)hipFree(s->v.acc);
    if (s->v.acc_slot) (void)hipFree(s->v.acc_slot);
    if (s->v.rank_agg) (void)hipFree(s->v.rank_agg);
    if (s->v.counters) (void)hipFree(s->v.counters);
    if (s->v.centres) (void)hipFree(s->v.centres);
    if (s->v.occ) (void)hipFree(s->v.occ);
    if (s->v.occ_delta) (void)hipFree(s->v.occ_delta);
    if (s->v.occ_touched) (void)hipFree(s->v.occ_touched);
    if (s->v.cells_alt) (void)hipFree(s->v.cells_alt);
    if (s->v.cell_sel) (void)hipFree(s->v.cell_sel);
    if (s->stage) (void)hipFree(s->stage);
    if (s->work) (void)hipFree(s->work);
    if (s->coop_work) (void)hipFree(s->coop_work);
    if (s->work_ev) (void)hipEventDestroy(s->work_ev);
    if (s->origins_dev) (void)hipFree(s->origins_dev);
    if (s->origins_ev) (void)hipEventDestroy(s->origins_ev);
    for (int k = 0; k < 4; k++)
        if (s->ev[k]) (void)hipEventDestroy(s->ev[k]);
    delete s;
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_mapset_set_centre(ndtgpu_mapset *s, size_t map, const double centre[3])
{
    if (!s || !centre || map >= s->n_maps) return fail(NDTGPU_ERR_INVALID, "set_centre: bad argument");
    for (int a = 0; a < 3; a++) s->centres_host[map * 3 + a] = centre[a];
    s->nice_host[map] = ndt_grid_is_nice(s->v.grid, centre) ? 1 : 0;
    HIP_TRY(hipMemcpy(s->v.centres + map * 3, centre, 3 * sizeof(double), hipMemcpyHostToDevice));
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_mapset_info(const ndtgpu_mapset *s, size_t *n_maps, int32_t cells_per_axis[3], uint32_t *max_cells)
{
    if (!s) return fail(NDTGPU_ERR_INVALID, "mapset_info: null");
    if (n_maps) *n_maps = s->n_maps;
    if (cells_per_axis)
        for (int a = 0; a < 3; a++) cells_per_axis[a] = s->v.grid.size[a];
    if (max_cells) *max_cells = s->v.grid.max_cells;
    return NDTGPU_OK;
}

// the build proper; `orig_dev`: per-map range origins already in device memory (or NULL: the grid centres)
static ndtgpu_status mapset_build_core(ndtgpu_mapset *s, size_t first, size_t count, const void *xyz_dev, size_t n_points,
                                       size_t stride_bytes, size_t map_stride_bytes, double range_limit,
                                       const double *orig_dev, const ndtgpu_cell_params *cell, hipStream_t st)
{
    if (!s || first + count > s->n_maps || (!xyz_dev && n_points) || stride_bytes < 12 || (stride_bytes & 3) ||
        n_points > 0xFFFFFFFFull)
        return fail(NDTGPU_ERR_INVALID, "mapset_build: bad argument");
    ndtgpu_cell_params cp;
    ndtgpu_default_cell_params(&cp);
    if (cell) cp = *cell;
    if (s->v.occ && count)   // a rebuilt map starts from cells without readings
        HIP_TRY(hipMemsetAsync(s->v.occ + first * (size_t)s->v.grid.slots, 0, count * (size_t)s->v.grid.slots * sizeof(float), st));
    if (s->profiling && !s->profile_span) HIP_TRY(hipEventRecord(s->ev[0], st));
    hipError_t e = ndt_launch_build(s->v, first, count, xyz_dev, n_points, stride_bytes, map_stride_bytes, range_limit,
                         orig_dev, cp.n_min, cp.eval_factor, s->nice_range(first, count), st);
    if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "mapset_build: launch", e);
    if (s->profiling && !s->profile_span) { HIP_TRY(hipEventRecord(s->ev[1], st)); s->ev_valid[0] = true; }
    return s->touch(st);
}

// per-map range origins from host memory into the set's buffer, ordered behind the last launch that read it
static ndtgpu_status upload_origins(ndtgpu_mapset *s, const double *range_origins, size_t count, hipStream_t st, const double **orig_dev)
{
    *orig_dev = nullptr;
    if (!range_origins || !count) return NDTGPU_OK;
    ndtgpu_status orc = s->origins_reserve(count * 3, st);
    if (orc != NDTGPU_OK) return orc;
    HIP_TRY(hipMemcpyAsync(s->origins_dev, range_origins, count * 3 * sizeof(double), hipMemcpyHostToDevice, st));
    *orig_dev = s->origins_dev;
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_mapset_build(ndtgpu_mapset *s, size_t first, size_t count, const void *xyz_dev, size_t n_points,
                                  size_t stride_bytes, size_t map_stride_bytes, double range_limit,
                                  const double *range_origins, const ndtgpu_cell_params *cell, ndtgpu_stream stream)
{
    if (!s || first + count > s->n_maps) return fail(NDTGPU_ERR_INVALID, "mapset_build: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const double *orig_dev = nullptr;
    ndtgpu_status rc = upload_origins(s, range_origins, count, st, &orig_dev);
    if (rc != NDTGPU_OK) return rc;
    rc = mapset_build_core(s, first, count, xyz_dev, n_points, stride_bytes, map_stride_bytes, range_limit, orig_dev, cell, st);
    if (orig_dev) { const ndtgpu_status urc = s->origins_used(st); if (rc == NDTGPU_OK) rc = urc; }
    return rc;
}

ndtgpu_status ndtgpu_profiling_enable(ndtgpu_mapset *s, int on)
{
    if (!s) return fail(NDTGPU_ERR_INVALID, "profiling_enable: null");
    if (on)
        for (int k = 0; k < 4; k++)
            if (!s->ev[k]) HIP_TRY(hipEventCreate(&s->ev[k]));
    s->profiling = on != 0;
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_last_kernel_ms(ndtgpu_mapset *s, int which, float *ms)
{
    if (!s || !ms || which < 0 || which > 1 || !s->ev_valid[which])
        return fail(NDTGPU_ERR_INVALID, "last_kernel_ms: nothing recorded");
    HIP_TRY(hipEventSynchronize(s->ev[2 * which + 1]));
    HIP_TRY(hipEventElapsedTime(ms, s->ev[2 * which], s->ev[2 * which + 1]));
    return NDTGPU_OK;
}

// Host clouds -> device staging area -> per-chunk kernel launches.  `count` clouds of n_points records, map_stride bytes
// apart, are cut into chunks of whole clouds that fit a pinned slot (16 MB); worker threads copy chunk after chunk from the
// caller's (pageable) memory into the ring of pinned slots, the calling thread sends every filled slot to the device on the
// copy stream and launches `launch(first_cloud, n_clouds, device pointer)` on `st` behind that copy: the copy of chunk
// k + 1 runs under the kernels of chunk k, nothing waits for the whole device.  Returns when the caller's memory has been
// read completely (the kernels may still be running on `st`).  Small inputs (< 24 MB) skip the ring: one ordinary copy.
static ndtgpu_status stage_host_clouds(ndtgpu_mapset *s, const void *xyz_host, size_t count, size_t n_points, size_t stride_bytes,
                                       size_t map_stride_bytes, hipStream_t st,
                                       const std::function<ndtgpu_status(size_t, size_t, const void *)> &launch)
{
    if (count == 0) return NDTGPU_OK;
    const size_t cloud_bytes = n_points * stride_bytes;
    size_t bytes = (count - 1) * map_stride_bytes + cloud_bytes;
    if (bytes == 0) bytes = 16;
    ndtgpu_status rc = s->ensure_stage(bytes);
    if (rc != NDTGPU_OK) return rc;
    // the staging area may still be read by the kernels of the previous call (another stream): order behind them
    if (s->stage_free_valid) HIP_TRY(hipStreamWaitEvent(st, s->stage_free_ev, 0));
    const size_t slot = ndtgpu_mapset::HOST_SLOT_BYTES;
    const char *force = getenv("NDTGPU_HOST_PIPE");            // 0: never the ring, 1: always (tests)
    // (the ring cuts the input into chunks of WHOLE clouds that lie one after the other: clouds that overlap in memory --
    //  map_stride_bytes < cloud_bytes, e.g. sliding windows over one point stream, or stride 0 -- take the single copy)
    const bool ring = n_points && map_stride_bytes >= cloud_bytes && map_stride_bytes <= slot && cloud_bytes <= slot &&
                      (force ? atoi(force) != 0 : bytes >= (24u << 20));
    if (!ring) {
        if (n_points) HIP_TRY(hipMemcpyAsync(s->stage, xyz_host, bytes, hipMemcpyHostToDevice, st));   // (pageable: returns when read)
        rc = launch(0, count, s->stage);
    } else {
        rc = s->ensure_host_ring();
        if (rc != NDTGPU_OK) return rc;
        if (s->stage_free_valid) HIP_TRY(hipStreamWaitEvent(s->host_copy_stream, s->stage_free_ev, 0));
        // clouds per chunk: a chunk of `per` clouds occupies (per - 1) * map_stride_bytes + cloud_bytes of its slot
        const size_t per = std::max<size_t>(1, std::min(count, 1 + (slot - cloud_bytes) / map_stride_bytes));
        const size_t n_chunks = (count + per - 1) / per;
        constexpr int R = ndtgpu_mapset::HOST_SLOTS;
        const int n_workers = (int)std::min<size_t>(4, n_chunks);
        std::vector<std::atomic<int>> staged(n_chunks);
        for (auto &a : staged) a.store(0, std::memory_order_relaxed);
        std::atomic<long> allowed((long)std::min<size_t>(n_chunks, R));     // chunks < allowed may be written into their slots
        std::atomic<int> stop(0);
        auto chunk_bytes = [&](size_t c) {
            const size_t c0 = c * per, cnt = std::min(per, count - c0);
            return (cnt - 1) * map_stride_bytes + cloud_bytes;
        };
        // a slot that an earlier CALL sent off may still be in flight
        for (int k = 0; k < R; k++)
            if (s->host_ev_used[k]) { HIP_TRY(hipEventSynchronize(s->host_ev[k])); s->host_ev_used[k] = false; }
        std::vector<std::thread> workers;
        for (int w = 0; w < n_workers; w++)
            workers.emplace_back([&, w]() {
                for (size_t c = (size_t)w; c < n_chunks && !stop.load(std::memory_order_relaxed); c += (size_t)n_workers) {
                    while ((long)c >= allowed.load(std::memory_order_acquire) && !stop.load(std::memory_order_relaxed)) std::this_thread::yield();
                    if (stop.load(std::memory_order_relaxed)) break;
                    memcpy(s->host_ring[c % R], (const char *)xyz_host + c * per * map_stride_bytes, chunk_bytes(c));
                    staged[c].store(1, std::memory_order_release);
                }
            });
        hipError_t herr = hipSuccess;
        ndtgpu_status lrc = NDTGPU_OK;
        // (profiling: one bracket around all chunks' launches -- ndtgpu_last_kernel_ms then reports the whole build, copies
        //  that the chunks wait for included -- instead of the last chunk's alone)
        const bool span = s->profiling && s->ev[0] && s->ev[1];
        if (span) { herr = hipEventRecord(s->ev[0], st); s->profile_span = true; }
        for (size_t c = 0; c < n_chunks && herr == hipSuccess && lrc == NDTGPU_OK; c++) {
            while (!staged[c].load(std::memory_order_acquire)) std::this_thread::yield();
            const int k = (int)(c % R);
            char *dst = (char *)s->stage + c * per * map_stride_bytes;
            herr = hipMemcpyAsync(dst, s->host_ring[k], chunk_bytes(c), hipMemcpyHostToDevice, s->host_copy_stream);
            if (herr == hipSuccess) herr = hipEventRecord(s->host_ev[k], s->host_copy_stream);
            if (herr == hipSuccess) { s->host_ev_used[k] = true; herr = hipStreamWaitEvent(st, s->host_ev[k], 0); }
            if (herr == hipSuccess) lrc = launch(c * per, std::min(per, count - c * per), dst);
            // chunk c + R will reuse this slot: it may be written once this copy has left the host.  (The calling thread
            // waits here while the workers fill the other slots and the device builds chunk c under the next copies.)
            if (c + R < n_chunks && herr == hipSuccess) {
                herr = hipEventSynchronize(s->host_ev[k]);
                s->host_ev_used[k] = false;
                allowed.store((long)(c + R) + 1, std::memory_order_release);
            }
        }
        if (span) {
            s->profile_span = false;
            if (herr == hipSuccess && lrc == NDTGPU_OK && hipEventRecord(s->ev[1], st) == hipSuccess) s->ev_valid[0] = true;
        }
        if (herr != hipSuccess || lrc != NDTGPU_OK) stop.store(1);
        for (auto &t : workers) t.join();
        rc = herr != hipSuccess ? fail(NDTGPU_ERR_HIP, "host clouds: staging copy", herr) : lrc;
    }
    // (also after a failure: chunks that were launched before it may still be reading the staging area)
    if (!s->stage_free_ev && hipEventCreateWithFlags(&s->stage_free_ev, hipEventDisableTiming) != hipSuccess) s->stage_free_ev = nullptr;
    if (s->stage_free_ev && hipEventRecord(s->stage_free_ev, st) == hipSuccess) s->stage_free_valid = true;
    else if (rc == NDTGPU_OK) return fail(NDTGPU_ERR_HIP, "host clouds: event");
    return rc;
}

ndtgpu_status ndtgpu_mapset_build_host_async(ndtgpu_mapset *s, size_t first, size_t count, const void *xyz_host,
                                             size_t n_points, size_t stride_bytes, size_t map_stride_bytes,
                                             double range_limit, const double *range_origins, const ndtgpu_cell_params *cell,
                                             ndtgpu_stream stream)
{
    if (!s || (!xyz_host && n_points) || count == 0 || first + count > s->n_maps)
        return fail(NDTGPU_ERR_INVALID, "mapset_build_host: bad argument");
    // (the range origins of ALL maps go up once, before the first chunk: a copy from pageable memory per chunk made the calling
    //  thread wait behind the previous chunk's build every time)
    hipStream_t st = (hipStream_t)stream;
    const double *orig_dev = nullptr;
    ndtgpu_status rc = upload_origins(s, range_origins, count, st, &orig_dev);
    if (rc != NDTGPU_OK) return rc;
    rc = stage_host_clouds(s, xyz_host, count, n_points, stride_bytes, map_stride_bytes, st,
                           [&](size_t c0, size_t cnt, const void *dev) {
                               return mapset_build_core(s, first + c0, cnt, dev, n_points, stride_bytes, map_stride_bytes,
                                                        range_limit, orig_dev ? orig_dev + 3 * c0 : nullptr, cell, st);
                           });
    if (orig_dev) { const ndtgpu_status urc = s->origins_used(st); if (rc == NDTGPU_OK) rc = urc; }
    return rc;
}

ndtgpu_status ndtgpu_mapset_build_host(ndtgpu_mapset *s, size_t first, size_t count, const void *xyz_host,
                                       size_t n_points, size_t stride_bytes, size_t map_stride_bytes,
                                       double range_limit, const double *range_origins, const ndtgpu_cell_params *cell)
{
    if (!s) return fail(NDTGPU_ERR_INVALID, "mapset_build_host: bad argument");
    ndtgpu_status rc = s->ensure_host_build_stream();
    if (rc != NDTGPU_OK) return rc;
    { ndtgpu_status wrc_ = s->wait_all(); if (wrc_ != NDTGPU_OK) return wrc_; }             // (earlier work on these maps, whatever stream it used)
    rc = ndtgpu_mapset_build_host_async(s, first, count, xyz_host, n_points, stride_bytes, map_stride_bytes, range_limit,
                                        range_origins, cell, (ndtgpu_stream)s->host_build_stream);
    if (rc != NDTGPU_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s->host_build_stream));       // the stream the maps were built on: no device-wide wait
    return NDTGPU_OK;
}

static ndtgpu_status read_counters(ndtgpu_mapset *s, size_t map, NdtMapCounters *c)
{
    { ndtgpu_status wrc_ = s->wait_all(); if (wrc_ != NDTGPU_OK) return wrc_; }
    HIP_TRY(hipMemcpy(c, s->v.counters + map, sizeof *c, hipMemcpyDeviceToHost));
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_mapset_counters(ndtgpu_mapset *s, size_t map, uint32_t out[8])
{
    if (!s || !out || map >= s->n_maps) return fail(NDTGPU_ERR_INVALID, "counters: bad argument");
    static_assert(sizeof(NdtMapCounters) == 32, "counter layout");
    NdtMapCounters c;
    ndtgpu_status rc = read_counters(s, map, &c);
    if (rc != NDTGPU_OK) return rc;
    memcpy(out, &c, 32);
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_mapset_num_cells(ndtgpu_mapset *s, size_t map, uint32_t *n)
{
    if (!s || !n || map >= s->n_maps) return fail(NDTGPU_ERR_INVALID, "num_cells: bad argument");
    NdtMapCounters c;
    ndtgpu_status rc = read_counters(s, map, &c);
    if (rc != NDTGPU_OK) return rc;
    *n = c.n_cells;
    if (c.overflow) return fail(NDTGPU_ERR_CAPACITY, "map needs more cells than max_cells");
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_mapset_export_cells(ndtgpu_mapset *s, size_t map, double *mean3, double *cov9, int32_t *idx3,
                                         uint32_t *npts)
{
    if (!s || map >= s->n_maps) return fail(NDTGPU_ERR_INVALID, "export_cells: bad argument");
    NdtMapCounters c;
    ndtgpu_status rc = read_counters(s, map, &c);
    if (rc != NDTGPU_OK) return rc;
    std::vector<NdtCell> host(c.n_cells);
    uint32_t sel = 0;
    if (s->v.cell_sel) HIP_TRY(hipMemcpy(&sel, s->v.cell_sel + map, sizeof sel, hipMemcpyDeviceToHost));
    if (c.n_cells)
        HIP_TRY(hipMemcpy(host.data(), ndt_cells_of(s->v, map, sel), c.n_cells * sizeof(NdtCell), hipMemcpyDeviceToHost));
    const NdtGrid &g = s->v.grid;
    for (uint32_t i = 0; i < c.n_cells; i++) {
        const NdtCell &k = host[i];
        if (mean3)
            for (int a = 0; a < 3; a++) mean3[3 * i + a] = k.mean[a];
        if (cov9) {
            double *o = cov9 + 9 * i;
            o[0] = k.cov[0]; o[1] = k.cov[1]; o[2] = k.cov[2];
            o[3] = k.cov[1]; o[4] = k.cov[3]; o[5] = k.cov[4];
            o[6] = k.cov[2]; o[7] = k.cov[4]; o[8] = k.cov[5];
        }
        if (idx3) {
            idx3[3 * i + 2] = (int32_t)(k.slot % g.size[2]);
            idx3[3 * i + 1] = (int32_t)((k.slot / g.size[2]) % g.size[1]);
            idx3[3 * i + 0] = (int32_t)(k.slot / ((uint32_t)g.size[2] * g.size[1]));
        }
        if (npts) npts[i] = k.n;
    }
    if (c.overflow) return fail(NDTGPU_ERR_CAPACITY, "map needs more cells than max_cells");
    return NDTGPU_OK;
}

// host-side packing of caller-provided Gaussians into NdtCell records keyed by LazyGrid slot
static ndtgpu_status pack_cells(const NdtGrid &g, const double *centre, const double *mean3, const double *cov9,
                                size_t n, bool need_slot, std::vector<NdtCell> &out)
{
    out.clear();
    out.reserve(n);
    for (size_t i = 0; i < n; i++) {
        NdtCell c;
        for (int a = 0; a < 3; a++) c.mean[a] = mean3[3 * i + a];
        const double *v = cov9 + 9 * i;
        c.cov[0] = v[0]; c.cov[1] = v[1]; c.cov[2] = v[2]; c.cov[3] = v[4]; c.cov[4] = v[5]; c.cov[5] = v[8];
        c.n = 1;
        c.slot = 0;
        if (need_slot) {
            int idx[3];
            bool inside = true;
            for (int a = 0; a < 3; a++) {
                idx[a] = lazygrid_index(c.mean[a], centre[a], g.res, g.size[a]);
                inside = inside && idx[a] >= 0 && idx[a] < g.size[a];
            }
            if (!inside) continue;   // LazyGrid::addPoint drops what falls outside the grid
            c.slot = (uint32_t)((idx[0] * g.size[1] + idx[1]) * g.size[2] + idx[2]);
        }
        out.push_back(c);
    }
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_mapset_set_cells(ndtgpu_mapset *s, size_t map, const double *mean3, const double *cov9,
                                      size_t n_cells)
{
    if (!s || map >= s->n_maps || (n_cells && (!mean3 || !cov9)))
        return fail(NDTGPU_ERR_INVALID, "set_cells: bad argument");
    std::vector<NdtCell> cells;
    pack_cells(s->v.grid, &s->centres_host[map * 3], mean3, cov9, n_cells, true, cells);
    // one Gaussian per slot (a later cell replaces an earlier one, like setMean/setCov on the same
    // NDTCell), sorted by slot = the canonical cell order of the build kernel
    std::stable_sort(cells.begin(), cells.end(), [](const NdtCell &a, const NdtCell &b) { return a.slot < b.slot; });
    std::vector<NdtCell> uniq;
    for (size_t i = 0; i < cells.size(); i++) {
        if (!uniq.empty() && uniq.back().slot == cells[i].slot) uniq.back() = cells[i];
        else uniq.push_back(cells[i]);
    }
    if (uniq.size() > s->v.grid.max_cells) return fail(NDTGPU_ERR_CAPACITY, "set_cells: more cells than max_cells");
    ndtgpu_status rc = s->ensure_stage(std::max<size_t>(uniq.size() * sizeof(NdtCell), 16));
    if (rc != NDTGPU_OK) return rc;
    if (!uniq.empty()) HIP_TRY(hipMemcpy(s->stage, uniq.data(), uniq.size() * sizeof(NdtCell), hipMemcpyHostToDevice));
    hipError_t e = ndt_launch_install_cells(s->v, map, (const NdtCell *)s->stage, uniq.size(), nullptr);
    if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "set_cells: launch", e);
    HIP_TRY(hipStreamSynchronize(nullptr));
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_derivatives(ndtgpu_mapset *t, size_t tmap, const double *src_mean3, const double *src_cov9,
                                 size_t m, int n_neighbours, int compute_hessian, double lfd1, double lfd2,
                                 double *score, double g[6], double H[36])
{
    if (!t || tmap >= t->n_maps || (m && (!src_mean3 || !src_cov9)) || !score || !g || n_neighbours < 0 ||
        n_neighbours > 3)
        return fail(NDTGPU_ERR_INVALID, "derivatives: bad argument");
    std::vector<NdtCell> cells;
    pack_cells(t->v.grid, nullptr, src_mean3, src_cov9, m, false, cells);
    size_t bytes = cells.size() * sizeof(NdtCell) + 32 * sizeof(double);
    ndtgpu_status rc = t->ensure_stage(bytes);
    if (rc != NDTGPU_OK) return rc;
    double *out_dev = (double *)t->stage;
    NdtCell *src_dev = (NdtCell *)((char *)t->stage + 32 * sizeof(double));
    { ndtgpu_status wrc_ = t->wait_all(); if (wrc_ != NDTGPU_OK) return wrc_; }
    if (!cells.empty()) HIP_TRY(hipMemcpy(src_dev, cells.data(), cells.size() * sizeof(NdtCell), hipMemcpyHostToDevice));
    hipError_t e = ndt_launch_derivatives(t->v, tmap, src_dev, cells.size(), n_neighbours, compute_hessian, lfd1, lfd2,
                                          out_dev, nullptr);
    if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "derivatives: launch", e);
    double out[28];
    HIP_TRY(hipMemcpy(out, out_dev, sizeof out, hipMemcpyDeviceToHost));
    *score = out[0];
    for (int a = 0; a < 6; a++) g[a] = out[1 + a];
    if (compute_hessian && H) {
        int o = 7;
        for (int a = 0; a < 6; a++)
            for (int b = a; b < 6; b++) { H[a * 6 + b] = out[o]; H[b * 6 + a] = out[o]; o++; }
    }
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_mapset_discard_cells(ndtgpu_mapset *s, size_t map, const float *xyz, size_t n_points)
{
    if (!s || map >= s->n_maps || (n_points && !xyz)) return fail(NDTGPU_ERR_INVALID, "discard_cells: bad argument");
    if (n_points == 0) return NDTGPU_OK;
    ndtgpu_status rc = s->ensure_stage(n_points * 3 * sizeof(float));
    if (rc != NDTGPU_OK) return rc;
    { ndtgpu_status wrc_ = s->wait_all(); if (wrc_ != NDTGPU_OK) return wrc_; }
    HIP_TRY(hipMemcpy(s->stage, xyz, n_points * 3 * sizeof(float), hipMemcpyHostToDevice));
    hipError_t e = ndt_launch_discard(s->v, map, (const float *)s->stage, n_points, nullptr);
    if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "discard_cells: launch", e);
    HIP_TRY(hipStreamSynchronize(nullptr));
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_mapset_enable_occupancy(ndtgpu_mapset *s)
{
    if (!s) return fail(NDTGPU_ERR_INVALID, "enable_occupancy: null");
    if (s->v.occ) return NDTGPU_OK;
    const size_t slots = (size_t)s->v.grid.slots, cap = s->v.grid.max_cells, n = s->n_maps;
    { ndtgpu_status wrc_ = s->wait_all(); if (wrc_ != NDTGPU_OK) return wrc_; }
    float *occ = nullptr;
    long long *delta = nullptr;
    NdtCell *alt = nullptr;
    uint32_t *sel = nullptr;
    unsigned char *touched = nullptr;
    const size_t blocks = (slots + 255) / 256;
    hipError_t e;
    if ((e = hipMalloc((void **)&occ, n * slots * sizeof(float))) != hipSuccess ||
        (e = hipMalloc((void **)&delta, n * slots * sizeof(long long))) != hipSuccess ||
        (e = hipMalloc((void **)&touched, n * blocks)) != hipSuccess ||
        (e = hipMemset(touched, 0, n * blocks)) != hipSuccess ||
        (e = hipMalloc((void **)&alt, n * cap * sizeof(NdtCell))) != hipSuccess ||
        (e = hipMalloc((void **)&sel, n * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMemset(occ, 0, n * slots * sizeof(float))) != hipSuccess ||
        (e = hipMemset(delta, 0, n * slots * sizeof(long long))) != hipSuccess ||
        (e = hipMemset(sel, 0, n * sizeof(uint32_t))) != hipSuccess) {
        if (occ) (void)hipFree(occ);
        if (delta) (void)hipFree(delta);
        if (touched) (void)hipFree(touched);
        if (alt) (void)hipFree(alt);
        if (sel) (void)hipFree(sel);
        return fail(NDTGPU_ERR_ALLOC, "enable_occupancy: device memory", e);
    }
    s->v.occ = occ; s->v.occ_delta = delta; s->v.occ_touched = touched; s->v.cells_alt = alt; s->v.cell_sel = sel;
    return NDTGPU_OK;
}

void ndtgpu_default_fuse_params(ndtgpu_fuse_params *p)
{
    p->maxz = 25.0;             // fuser_hmt.cpp:485
    p->sensor_noise = 0.06;
    p->maxnumpoints = 1e5;      // fuser_hmt.cpp:486
    p->occupancy_limit = 255.0;
    p->n_min = 3;
    p->eval_factor = 1000.0;
}

ndtgpu_status ndtgpu_mapset_add_cloud(ndtgpu_mapset *s, size_t first, size_t count, const void *xyz_dev, size_t n_points,
                                      size_t stride_bytes, size_t map_stride_bytes, const double *origins,
                                      const ndtgpu_fuse_params *prm, ndtgpu_stream stream)
{
    if (!s || first + count > s->n_maps || (!xyz_dev && n_points) || stride_bytes < 12 || (stride_bytes & 3) ||
        n_points > 0xFFFFFFFFull || (count && !origins))
        return fail(NDTGPU_ERR_INVALID, "add_cloud: bad argument");
    if (!s->v.occ) return fail(NDTGPU_ERR_INVALID, "add_cloud: call ndtgpu_mapset_enable_occupancy first (NDTMap::initialize)");
    if (count == 0) return NDTGPU_OK;
    ndtgpu_fuse_params fp;
    ndtgpu_default_fuse_params(&fp);
    if (prm) fp = *prm;
    if (!(fp.occupancy_limit > 0)) return fail(NDTGPU_ERR_INVALID, "add_cloud: occupancy_limit must be positive");
    hipStream_t st = (hipStream_t)stream;
    {
        ndtgpu_status orc = s->origins_reserve(count * 3, st);
        if (orc != NDTGPU_OK) return orc;
    }
    HIP_TRY(hipMemcpyAsync(s->origins_dev, origins, count * 3 * sizeof(double), hipMemcpyHostToDevice, st));
    NdtFuseParams p;
    p.maxz = fp.maxz; p.sensor_noise = fp.sensor_noise; p.maxnumpoints = fp.maxnumpoints;
    p.occupancy_limit = fp.occupancy_limit; p.eval_factor = fp.eval_factor; p.n_min = fp.n_min;
    hipError_t e = ndt_launch_fuse(s->v, first, count, xyz_dev, n_points, stride_bytes, map_stride_bytes, s->origins_dev, p,
                                   s->nice_range(first, count), st);
    if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "add_cloud: launch", e);
    { ndtgpu_status trc = s->touch(st); if (trc != NDTGPU_OK) return trc; }
    return s->origins_used(st);
}

ndtgpu_status ndtgpu_mapset_add_cloud_host_async(ndtgpu_mapset *s, size_t first, size_t count, const void *xyz_host,
                                                 size_t n_points, size_t stride_bytes, size_t map_stride_bytes,
                                                 const double *origins, const ndtgpu_fuse_params *prm, ndtgpu_stream stream)
{
    if (!s || (!xyz_host && n_points) || count == 0 || first + count > s->n_maps || !origins)
        return fail(NDTGPU_ERR_INVALID, "add_cloud_host: bad argument");
    return stage_host_clouds(s, xyz_host, count, n_points, stride_bytes, map_stride_bytes, (hipStream_t)stream,
                             [&](size_t c0, size_t cnt, const void *dev) {
                                 return ndtgpu_mapset_add_cloud(s, first + c0, cnt, dev, n_points, stride_bytes, map_stride_bytes,
                                                                origins + 3 * c0, prm, stream);
                             });
}

ndtgpu_status ndtgpu_mapset_add_cloud_host(ndtgpu_mapset *s, size_t first, size_t count, const void *xyz_host,
                                           size_t n_points, size_t stride_bytes, size_t map_stride_bytes,
                                           const double *origins, const ndtgpu_fuse_params *prm)
{
    if (!s) return fail(NDTGPU_ERR_INVALID, "add_cloud_host: bad argument");
    ndtgpu_status rc = s->ensure_host_build_stream();
    if (rc != NDTGPU_OK) return rc;
    { ndtgpu_status wrc_ = s->wait_all(); if (wrc_ != NDTGPU_OK) return wrc_; }
    rc = ndtgpu_mapset_add_cloud_host_async(s, first, count, xyz_host, n_points, stride_bytes, map_stride_bytes, origins, prm,
                                            (ndtgpu_stream)s->host_build_stream);
    if (rc != NDTGPU_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s->host_build_stream));
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_mapset_clear(ndtgpu_mapset *s, size_t first, size_t count)
{
    if (!s || first + count > s->n_maps) return fail(NDTGPU_ERR_INVALID, "mapset_clear: bad argument");
    if (count == 0) return NDTGPU_OK;
    { ndtgpu_status wrc_ = s->wait_all(); if (wrc_ != NDTGPU_OK) return wrc_; }
    const NdtGrid &g = s->v.grid;
    const size_t slots = (size_t)g.slots;
    HIP_TRY(hipMemset(s->v.rankmap + first * ndt_rm_stride(g), 0, count * ndt_rm_stride(g) * sizeof(uint2)));
    HIP_TRY(hipMemset(s->v.counters + first, 0, count * sizeof(NdtMapCounters)));
    if (s->v.occ) {
        HIP_TRY(hipMemset(s->v.occ + first * slots, 0, count * slots * sizeof(float)));
        HIP_TRY(hipMemset(s->v.cell_sel + first, 0, count * sizeof(uint32_t)));
    }
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_mapset_export_occupancy(ndtgpu_mapset *s, size_t map, float *occ_out)
{
    if (!s || map >= s->n_maps || !occ_out) return fail(NDTGPU_ERR_INVALID, "export_occupancy: bad argument");
    if (!s->v.occ) return fail(NDTGPU_ERR_INVALID, "export_occupancy: occupancy not enabled on this set");
    { ndtgpu_status wrc_ = s->wait_all(); if (wrc_ != NDTGPU_OK) return wrc_; }
    HIP_TRY(hipMemcpy(occ_out, s->v.occ + map * (size_t)s->v.grid.slots, (size_t)s->v.grid.slots * sizeof(float), hipMemcpyDeviceToHost));
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_mapset_import_occupancy(ndtgpu_mapset *s, size_t map, const float *occ)
{
    if (!s || map >= s->n_maps || !occ) return fail(NDTGPU_ERR_INVALID, "import_occupancy: bad argument");
    if (!s->v.occ) return fail(NDTGPU_ERR_INVALID, "import_occupancy: occupancy not enabled on this set");
    { ndtgpu_status wrc_ = s->wait_all(); if (wrc_ != NDTGPU_OK) return wrc_; }
    HIP_TRY(hipMemcpy(s->v.occ + map * (size_t)s->v.grid.slots, occ, (size_t)s->v.grid.slots * sizeof(float), hipMemcpyHostToDevice));
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_overlap_score_batch(ndtgpu_mapset *rs, const uint32_t *ridx, ndtgpu_mapset *ms, const uint32_t *midx,
                                         const double *T16, size_t n_links, double *score, int64_t *nb_sum,
                                         ndtgpu_stream stream)
{
    if (!rs || !ms || (n_links && (!ridx || !midx || !T16 || !score)))
        return fail(NDTGPU_ERR_INVALID, "overlap_score: bad argument");
    if (!rs->v.occ || !ms->v.occ) return fail(NDTGPU_ERR_INVALID, "overlap_score: occupancy not enabled on both sets");
    if (n_links == 0) return NDTGPU_OK;
    for (size_t k = 0; k < n_links; k++)
        if (ridx[k] >= rs->n_maps || midx[k] >= ms->n_maps) return fail(NDTGPU_ERR_INVALID, "overlap_score: map index");
    hipStream_t st = (hipStream_t)stream;
    { ndtgpu_status wrc_ = rs->wait_all(); if (wrc_ != NDTGPU_OK) return wrc_; }
    { ndtgpu_status wrc_ = ms->wait_all(); if (wrc_ != NDTGPU_OK) return wrc_; }
    // Every link walks a LIST of its moving map's cells with a reading: every distinct moving map of the batch is scanned twice
    // (count, then its (slot, occupancy) pairs in slot order), a link then visits a few thousand pairs instead of the map's whole
    // dense array.  Which kernel serves a link does NOT depend on the other links of its batch (the dense kernel adds in
    // another order: a link's score must not change with the way links are chunked or sharded over ranks).
    // NDTGPU_OVERLAP_DENSE=1: the dense kernel for every link (A/B).
    const char *dense_env = getenv("NDTGPU_OVERLAP_DENSE");
    std::vector<uint32_t> list_of_link, list_maps;
    if (!(dense_env && atoi(dense_env) != 0)) {
        std::vector<int32_t> list_of_map(ms->n_maps, -1);
        list_of_link.resize(n_links);
        for (size_t k = 0; k < n_links; k++) {
            int32_t &u = list_of_map[midx[k]];
            if (u < 0) { u = (int32_t)list_maps.size(); list_maps.push_back(midx[k]); }
            list_of_link[k] = (uint32_t)u;
        }
    }
    const size_t U = list_maps.size();
    const size_t bT = n_links * 16 * sizeof(double), bI = n_links * sizeof(uint32_t), bS = n_links * sizeof(double);
    const size_t off_r = (bT + 255) & ~(size_t)255, off_m = (off_r + bI + 255) & ~(size_t)255,
                 off_s = (off_m + bI + 255) & ~(size_t)255, off_n = (off_s + bS + 255) & ~(size_t)255,
                 off_l = (off_n + n_links * sizeof(long long) + 255) & ~(size_t)255,          // list of every link
                 off_u = (off_l + (U ? bI : 0) + 255) & ~(size_t)255,                          // the maps of the lists
                 off_o = (off_u + U * sizeof(uint32_t) + 255) & ~(size_t)255,                  // counts, then offsets
                 off_p = (off_o + (U + 1) * sizeof(unsigned) + 255) & ~(size_t)255;            // the pairs
    ndtgpu_status rc = rs->ensure_stage(off_p);
    if (rc != NDTGPU_OK) return rc;
    std::vector<unsigned> offs(U + 1, 0u);
    if (U) {
        // pass 1: how many cells with a reading every listed map has (before the buffer is laid out for good: it may move)
        char *b0 = (char *)rs->stage;
        HIP_TRY(hipMemcpyAsync(b0 + off_u, list_maps.data(), U * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        hipError_t e0 = ndt_launch_occ_count(ms->v, 0, (const uint32_t *)(b0 + off_u), U, (unsigned *)(b0 + off_o), st);
        if (e0 != hipSuccess) return fail(NDTGPU_ERR_HIP, "overlap_score: count launch", e0);
        HIP_TRY(hipMemcpyAsync(offs.data(), b0 + off_o, U * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        unsigned long long at = 0;
        for (size_t u = 0; u < U; u++) { const unsigned c = offs[u]; offs[u] = (unsigned)at; at += c; }
        if (at > 0xFFFFFFFFull) return fail(NDTGPU_ERR_CAPACITY, "overlap_score: too many occupied cells in one batch");
        offs[U] = (unsigned)at;
        rc = rs->ensure_stage(off_p + (size_t)at * 8u);
        if (rc != NDTGPU_OK) return rc;
    }
    char *base = (char *)rs->stage;
    HIP_TRY(hipMemcpyAsync(base, T16, bT, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(base + off_r, ridx, bI, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(base + off_m, midx, bI, hipMemcpyHostToDevice, st));
    hipError_t e;
    if (U) {
        HIP_TRY(hipMemcpyAsync(base + off_l, list_of_link.data(), bI, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(base + off_u, list_maps.data(), U * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(base + off_o, offs.data(), (U + 1) * sizeof(unsigned), hipMemcpyHostToDevice, st));
        e = ndt_launch_occ_list(ms->v, (const uint32_t *)(base + off_u), U, (const unsigned *)(base + off_o), base + off_p, st);
        if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "overlap_score: list launch", e);
        e = ndt_launch_overlap_lists(rs->v, (const uint32_t *)(base + off_r), ms->v, (const uint32_t *)(base + off_m),
                                     (const uint32_t *)(base + off_l), (const unsigned *)(base + off_o), base + off_p,
                                     (const double *)base, n_links, (double *)(base + off_s), (long long *)(base + off_n), st);
    } else {
        e = ndt_launch_overlap(rs->v, (const uint32_t *)(base + off_r), ms->v, (const uint32_t *)(base + off_m),
                               (const double *)base, n_links, (double *)(base + off_s), (long long *)(base + off_n), st);
    }
    if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "overlap_score: launch", e);
    HIP_TRY(hipMemcpyAsync(score, base + off_s, bS, hipMemcpyDeviceToHost, st));
    if (nb_sum) HIP_TRY(hipMemcpyAsync(nb_sum, base + off_n, n_links * sizeof(long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return NDTGPU_OK;
}

static NdtMatchParamsDev to_dev(const ndtgpu_match_params *p)
{
    ndtgpu_match_params d;
    ndtgpu_default_match_params(&d);
    if (p) d = *p;
    NdtMatchParamsDev o;
    o.n_neighbours = d.n_neighbours;
    o.itr_max = d.itr_max;
    o.step_control = d.step_control;
    o.dof_mask = d.dof_mask;
    o.use_initial_guess = d.use_initial_guess;
    o.delta_score = d.delta_score;
    o.lfd1 = d.lfd1;
    o.lfd2 = d.lfd2;
    o.fusion_flags = 1;
    return o;
}

static_assert(sizeof(NdtMatchResultDev) == sizeof(ndtgpu_match_result), "result layouts must agree");

// ---- the grid-barrier matcher (csrc/ndt_match.hip ndt_match_coop_kernel): several workgroups per registration ---------
// One such launch at a time on the device: two of them could each hold part of the chip and wait for the rest.  Every
// launch waits (on its stream, not on the host) for the event of the one before it.
#define NDTGPU_HOST_LOOP_MAX 8             // up to this many registrations per call: the latency shapes (grid barrier / host loop)
#define NDTGPU_COOP_MIN_SET_CELLS 16384u   // source sets with room for fewer cells per map hold small (2D) maps
static std::mutex g_coop_mutex;
// (per device: an event belongs to the device it was created on, and launches on one device need not wait for another's)
#define NDTGPU_MAX_DEVICES 64
static hipEvent_t g_coop_ev_dev[NDTGPU_MAX_DEVICES] = {};
static bool g_coop_ev_valid_dev[NDTGPU_MAX_DEVICES] = {};
static int coop_dev()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
    return d % NDTGPU_MAX_DEVICES;
}
#define g_coop_ev g_coop_ev_dev[coop_dev()]
#define g_coop_ev_valid g_coop_ev_valid_dev[coop_dev()]

ndtgpu_status ndtgpu_mapset::ensure_coop(size_t bytes)
{
    if (bytes > coop_bytes && g_coop_ev_valid) HIP_TRY(hipEventSynchronize(g_coop_ev));   // an asynchronous launch may still use the area
    return ensure_coop_impl(bytes);
}

struct CoopPlan {
    unsigned groups, per_group;   // workgroups per registration in the grid (task pool: of the launch); source cells per chunk
    size_t stride;                // bytes of work area per registration
    int checked;                  // launch through hipLaunchCooperativeKernel
    bool pool;                    // the task-pool kernel (default) instead of the grid-barrier kernel (NDTGPU_POOL=0)
};

// The grid of a batch: every registration gets the same number of workgroups, as many as fit on the chip together
// (occupancy query), at most one per chunk of the largest map the source set can hold.  No look at the maps: the kernel
// cuts a registration into chunks by its own cell count and surplus workgroups leave at once.  False: the batch does not
// fit (more registrations than resident workgroups).
static bool coop_plan(const ndtgpu_mapset *ss, size_t n_pairs, const NdtMatchParamsDev &p, CoopPlan &pl)
{
    const unsigned capacity = ndt_match_coop_capacity(p.n_neighbours);
    if (capacity == 0 || n_pairs == 0 || n_pairs > capacity) return false;
    // Source cells per chunk -- a property of the source SET (its cell capacity), so that a registration's rows, and with
    // them its bits, do not depend on the batch it is in.  Sets of small maps (planar scans): 96 (12 per wave; with ONE grid
    // barrier per evaluation, round 6: 64 / 96 / 128 / 192 cells 0.295 / 0.282 / 0.297 / 0.288 ms for the 2D pair of 100 k
    // points, build included).  Sets that hold large maps (3D sweeps, >= 16 k cells): 256 -- every chunk
    // costs its own pass over the pair terms (batches of 64 that end half empty) and its own wave sum, and every row a
    // hand-over: 12 k-cell maps, 4 / 8 / 16 / 32 pairs 1.41 / 2.33 / 3.57 / 4.75 ms with 128 against 1.34 / 2.20 / 3.29 /
    // 4.34 ms with 256 (384: 32 pairs 4.63, 512: 4.56); one pair alone 1.11 against 1.33 ms -- half as many workgroups.
    // (Round 6, one barrier per evaluation: one pair alone 1.04 / 1.17 / 1.26 ms with 96 / 192 / 256, but the pool's 32 pairs
    //  5.07 / 4.83 / 4.63 ms: the batch decides, 256 stays.)
    const char *cpg = getenv("NDTGPU_COOP_CELLS");
    pl.per_group = (cpg && atoi(cpg) > 0) ? (unsigned)atoi(cpg) : (ss->v.grid.max_cells >= 16384u ? 256u : 96u);
    const unsigned n_chunks = std::max(1u, (ss->v.grid.max_cells + pl.per_group - 1u) / pl.per_group);
    // Up to 8 registrations: the grid-barrier kernel (static teams, the solver state stays in one workgroup's LDS: 12 k-cell
    // 3D maps, 1 / 4 / 8 pairs 1.63 / 2.30 / 3.01 ms against 1.94 / 2.57 / 3.14 ms).  More: the task pool (any workgroup
    // takes any task of any registration, the long registrations get the workgroups the others leave: 16 / 32 pairs
    // 5.6 / 6.9 ms against 7.2 / 10.7 ms).  Same chunks, same order of sums: the same bits.  NDTGPU_POOL=0 / 1 forces one.
    const char *pool_env = getenv("NDTGPU_POOL");
    pl.pool = pool_env ? atoi(pool_env) != 0 : n_pairs > NDTGPU_HOST_LOOP_MAX;
    if (pl.pool) {
        // any workgroup takes any task of any registration: as many workgroups as the chip holds, or as there can be tasks
        pl.groups = (unsigned)std::max<size_t>(1, std::min<size_t>(capacity, n_pairs * (size_t)n_chunks));
        const char *pg = getenv("NDTGPU_POOL_GROUPS");                        // (experiments: workgroups of the launch)
        if (pg && atoi(pg) > 0) pl.groups = (unsigned)atoi(pg);
        pl.stride = ndt_match_pool_pair_bytes(n_chunks);
    } else {
        pl.groups = std::max<unsigned>(1u, std::min<size_t>(n_chunks, capacity / n_pairs));
        pl.stride = ndt_match_coop_work_bytes(n_chunks);
    }
    const char *api_env = getenv("NDTGPU_COOP_API");          // NDTGPU_COOP_API=1: hipLaunchCooperativeKernel (checked by the runtime)
    pl.checked = (api_env && atoi(api_env) != 0) ? 1 : 0;
    return true;
}

// Enqueues ONE launch for the whole batch on `st` behind the previous grid-barrier launch of the process (g_coop_mutex held).
static ndtgpu_status coop_enqueue(ndtgpu_mapset *ts, ndtgpu_mapset *ss, const uint32_t *tidx_dev, const uint32_t *sidx_dev,
                                  double *T16_dev, NdtMatchResultDev *res_dev, const double *Q36_dev, size_t n_pairs,
                                  const NdtMatchParamsDev &p, const CoopPlan &pl, bool clear, bool record, hipStream_t st,
                                  unsigned *done_host = nullptr)
{
    if (g_coop_ev_valid) HIP_TRY(hipStreamWaitEvent(st, g_coop_ev, 0));
    // the control blocks must be zero (barrier counters only grow while a registration runs); the kernels leave them so
    hipError_t e;
    if (pl.pool) {
        if (clear) {
            HIP_TRY(hipMemsetAsync(ts->coop_work, 0, ndt_match_pool_ctrl_bytes(), st));
            HIP_TRY(hipMemset2DAsync((char *)ts->coop_work + ndt_match_pool_ctrl_bytes(), pl.stride, 0, ndt_match_pool_head_bytes(), n_pairs, st));
        }
        e = ndt_launch_match_pool(ts->v, tidx_dev, ss->v, sidx_dev, T16_dev, n_pairs, p, res_dev, Q36_dev, pl.groups, pl.per_group,
                                  ts->coop_work, pl.stride, st);
    } else {
        if (clear) HIP_TRY(hipMemset2DAsync(ts->coop_work, pl.stride, 0, ndt_match_coop_ctrl_bytes(), n_pairs, st));
        e = ndt_launch_match_coop(ts->v, tidx_dev, ss->v, sidx_dev, T16_dev, 0, n_pairs, p, res_dev, Q36_dev, pl.groups,
                                  pl.per_group, ts->coop_work, pl.stride, pl.checked, st, done_host);
    }
    if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "match: grid-barrier launch", e);
    if (record) {     // (a caller that waits for its launch under the mutex leaves nothing for later launches to wait for)
        if (!g_coop_ev) HIP_TRY(hipEventCreateWithFlags(&g_coop_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(g_coop_ev, st));
        g_coop_ev_valid = true;
    }
    { ndtgpu_status trc = ts->touch(st); if (trc != NDTGPU_OK) return trc; }
    if (ss != ts) { ndtgpu_status trc = ss->touch(st); if (trc != NDTGPU_OK) return trc; }
    return NDTGPU_OK;
}

// The persistent matcher on device-resident arguments: asynchronous on `stream`.
static ndtgpu_status match_device_core(ndtgpu_mapset *ts, const uint32_t *tidx_dev, ndtgpu_mapset *ss, const uint32_t *sidx_dev,
                                       double *T16_dev, size_t n_pairs, const NdtMatchParamsDev &p,
                                       ndtgpu_match_result *results_dev, const double *Q36_dev, hipStream_t st,
                                       const unsigned *feat_off_dev = nullptr, const double *feat_cells_dev = nullptr)
{
    if (n_pairs == 0) return NDTGPU_OK;
    // persistent workgroups, one per CU (8 waves x 256 VGPRs), each with `slots` registrations in flight whose evaluation
    // shares its waves take in turn (csrc/ndt_match.hip); pairs are pulled from a ticket counter.
    // NDTGPU_PARK_ITERS: iterations after which a long registration yields to a fresh pair.  NDTGPU_SLOTS=1: one
    // registration per workgroup (A/B; the results are the same bits).  NDTGPU_DOUBLE_THRESH: a workgroup resumes a
    // second parked registration only when more than this many are waiting.
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
        n_cu = 256;
    const char *park_env = getenv("NDTGPU_PARK_ITERS");       // read per call: tests switch it
    const int park_iters = park_env ? atoi(park_env) : 6;
    const char *slots_env = getenv("NDTGPU_SLOTS");
    const int slots = (slots_env && atoi(slots_env) >= 1 && atoi(slots_env) <= 3) ? atoi(slots_env) : 2;
    unsigned n_groups = (unsigned)std::min<size_t>(n_pairs, (size_t)n_cu);
    // (a stream that owns only part of the chip -- hipExtStreamCreateWithCUMask, bench.py --cu-split -- wants one workgroup
    //  per CU it has, not per CU of the device)
    if (ts->match_groups) n_groups = (unsigned)std::min<size_t>(n_pairs, (size_t)ts->match_groups);
    const char *grp_env = getenv("NDTGPU_MATCH_GROUPS");
    // (more workgroups than CUs: narrow-workgroup builds of the kernel, -DNDT_MATCH_THREADS=256, of which two share a CU)
    if (grp_env && atoi(grp_env) > 0) n_groups = (unsigned)std::min<size_t>(n_pairs, (size_t)atoi(grp_env));
    const char *dbl_env = getenv("NDTGPU_DOUBLE_THRESH");
    const unsigned double_thresh = dbl_env ? (unsigned)atoi(dbl_env) : n_groups;
    // The work area (ticket counters, parked solver states) belongs to the target set: a launch on another stream
    // waits for the previous one, and growing the area waits for everything that may still use the old one.
    if (ts->work_ev_valid && ts->work_stream != st) HIP_TRY(hipStreamWaitEvent(st, ts->work_ev, 0));
    const size_t need = ndt_match_work_bytes(n_pairs, (size_t)n_groups * slots);
    if (need > ts->work_bytes && ts->work_ev_valid) HIP_TRY(hipEventSynchronize(ts->work_ev));
    ndtgpu_status wrc = ts->ensure_work(need);
    if (wrc != NDTGPU_OK) return wrc;
    if (ts->profiling) HIP_TRY(hipEventRecord(ts->ev[2], st));
    hipError_t e = ndt_launch_match(ts->v, tidx_dev, ss->v, sidx_dev, T16_dev, n_pairs, p,
                                    reinterpret_cast<NdtMatchResultDev *>(results_dev), Q36_dev, feat_off_dev, feat_cells_dev,
                                    n_groups, park_iters, slots, double_thresh, ts->work, st);
    if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "match: launch", e);
    if (ts->profiling) { HIP_TRY(hipEventRecord(ts->ev[3], st)); ts->ev_valid[1] = true; }
    if (!ts->work_ev) HIP_TRY(hipEventCreateWithFlags(&ts->work_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ts->work_ev, st));
    ts->work_ev_valid = true;
    ts->work_stream = st;
    // the launch reads both sets' maps: host-synchronous rebuilds of either wait for it
    { ndtgpu_status trc = ts->touch(st); if (trc != NDTGPU_OK) return trc; }
    if (ss != ts) { ndtgpu_status trc = ss->touch(st); if (trc != NDTGPU_OK) return trc; }
    return NDTGPU_OK;
}

static ndtgpu_status match_coop(ndtgpu_mapset *ts, const uint32_t *tidx, ndtgpu_mapset *ss, const uint32_t *sidx,
                                double *T16, size_t n_pairs, const NdtMatchParamsDev &p, const double *Q36,
                                ndtgpu_match_result *results, hipStream_t st, bool *done);

ndtgpu_status ndtgpu_match_batch_device(ndtgpu_mapset *ts, const uint32_t *tidx_dev, ndtgpu_mapset *ss,
                                        const uint32_t *sidx_dev, double *T16_dev, size_t n_pairs,
                                        const ndtgpu_match_params *prm, ndtgpu_match_result *results_dev,
                                        ndtgpu_stream stream)
{
    if (!ts || !ss || (n_pairs && (!tidx_dev || !sidx_dev || !T16_dev || !results_dev)))
        return fail(NDTGPU_ERR_INVALID, "match_batch_device: bad argument");
    NdtMatchParamsDev p = to_dev(prm);
    p.fusion_flags = 0;
    if (p.n_neighbours < 0 || p.n_neighbours > 3 || (p.dof_mask & 0x3f) == 0)
        return fail(NDTGPU_ERR_INVALID, "match: n_neighbours must be 0..3 and dof_mask non-empty");
    // A batch that cannot fill the chip with one persistent workgroup per registration (at most half as many pairs as
    // CUs) on maps large enough to be split (a source set that holds >= 16 k cells per map): the grid-barrier matcher,
    // as many workgroups per registration as fit on the chip together -- ONE asynchronous launch that reads indices and
    // poses where they are, ordered behind the previous launch of its kind by an event.  NDTGPU_DEVICE_COOP=0 keeps such
    // batches on the persistent kernel; so does a stream that is being captured (the event is not part of the capture).
    {
        hipStream_t st = (hipStream_t)stream;
        int dev = 0, n_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
            n_cu = 256;
        const char *dc_env = getenv("NDTGPU_DEVICE_COOP");
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        CoopPlan pl;
        if (n_pairs > 0 && n_pairs <= (size_t)n_cu / 2 && ss->v.grid.max_cells >= NDTGPU_COOP_MIN_SET_CELLS && !(dc_env && atoi(dc_env) == 0) &&
            hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone && coop_plan(ss, n_pairs, p, pl)) {
            std::lock_guard<std::mutex> coop_lock(g_coop_mutex);
            ndtgpu_status rc = ts->ensure_coop(n_pairs * pl.stride + (pl.pool ? ndt_match_pool_ctrl_bytes() : 0));
            if (rc != NDTGPU_OK) return rc;
            ts->coop_clean_stride = pl.stride;
            ts->coop_clean_upto = 0;                            // (nobody will look how this launch ended: the next call clears)
            ts->ev_valid[1] = false;
            return coop_enqueue(ts, ss, tidx_dev, sidx_dev, T16_dev, reinterpret_cast<NdtMatchResultDev *>(results_dev), nullptr,
                                n_pairs, p, pl, true, true, st);
        }
    }
    return match_device_core(ts, tidx_dev, ss, sidx_dev, T16_dev, n_pairs, p, results_dev, nullptr, (hipStream_t)stream);
}

// ---- the registrar: scans in, poses out (include/ndtgpu.h) ---------------------------------------------------------------
// `depth` map sets with a stream each.  A sub-batch = ONE build launch for its 2 p scans + ONE matcher launch on the next
// stream in turn.  What orders the streams: (i) the inputs (an event recorded on the caller's stream at the call), (ii) the
// builds among themselves -- sub-batch k + 1 builds once sub-batch k's build has finished, i.e. while matcher k runs: the
// matcher's workgroups leave their CUs as soon as no registration is left to start (csrc/ndt_match.hip), so the next builds
// fill the CUs that the few long registrations do not hold --, (iii) a map set against its own previous use (same stream).
// the Gaussian cells of n_maps maps, summed, to two words of pinned host memory: {cells, seq + 1}
__global__ void ndt_reg_stats_kernel(const NdtMapCounters *ctr, unsigned n_maps, unsigned long long seq, unsigned long long *out)
{
    __shared__ unsigned long long part[256];
    unsigned long long c = 0;
    for (unsigned i = threadIdx.x; i < n_maps; i += 256u) c += ctr[i].n_cells;
    part[threadIdx.x] = c;
    __syncthreads();
    for (unsigned o = 128u; o > 0u; o >>= 1) {
        if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&out[0], part[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&out[1], seq + 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

struct ndtgpu_registrar {
    size_t per = 0;
    int depth = 0;
    ndtgpu_registrar_params prm{};     // as given to ndtgpu_registrar_create_ex (zeros resolved)
    std::vector<ndtgpu_mapset *> sets;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> built;
    // completion events, one per sub-batch, in a ring of 4 x depth: sub-batch j records done[j % ring] -- an entry always
    // belongs to stream j % depth, so whoever waits on an entry that a later sub-batch has re-recorded waits for a superset
    std::vector<hipEvent_t> done;
    hipEvent_t in_ev = nullptr;
    int last_built = -1;
    uint32_t *iota = nullptr;          // device: 0 .. 2 per - 1 (target indices: iota, source indices: iota + p)
    size_t submitted = 0;              // sub-batches so far
    // stream-fed form (csrc/ndt_match.hip, ndt_match_stream_kernel): build streams, ONE matcher stream on which an instance of
    // the matcher serves batch after batch from a queue in device memory
    void *queue = nullptr;
    hipStream_t bst = nullptr, bst2 = nullptr, pst = nullptr, mst = nullptr;   // builds (two, in turn), publishes (in order), matcher
    int mst_prio = 0;                  // the matcher stream's priority: a stream of the caller's at this priority may share its hardware queue
    hipStream_t hst = nullptr;         // the drain helper of ndtgpu_registrar_sync
    size_t helped = 0;                 // sub-batches submitted when the last helper was launched
    std::vector<hipEvent_t> trace_ev;  // NDTGPU_REG_TRACE: start / end of the build of the last 64 sub-batches
    size_t trace_first = (size_t)-1;   // ... the first sub-batch that has them
    std::vector<hipEvent_t> pub_ev;
    unsigned stream_groups = 0;        // workgroups (= CUs) of a matcher instance; 0: to be measured on the next sub-batch
    int stream_slots = 2;              // registrations in flight per workgroup of an instance (2, or 3 with half the hit list each)
    int device = 0;
    hipEvent_t probe_ev[2] = {nullptr, nullptr};
    std::vector<std::pair<unsigned, hipStream_t>> masked;      // streams that own the first F CUs of the mask (the split's build probes)
    ndtgpu_status masked_stream(unsigned n_cus, hipStream_t *out)
    {
        for (auto &m : masked) if (m.first == n_cus) { *out = m.second; return NDTGPU_OK; }
        std::vector<uint32_t> mask(((size_t)n_cu + 31) / 32, 0u);
        for (unsigned i = 0; i < n_cus && i < (unsigned)n_cu; i++) mask[i / 32] |= 1u << (i % 32);
        hipStream_t st = nullptr;
        HIP_TRY(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
        masked.emplace_back(n_cus, st);
        *out = st;
        return NDTGPU_OK;
    }
    int stream_nn = -1;
    int n_cu = 256;
    // the split of the chip is measured on a sub-batch and re-measured when the maps change: per slot the map counters of
    // the last build travel to pinned host memory on a side stream; a later call looks at what has arrived (no waiting)
    int calibrations = 0;
    double calib_cells = 0.0;          // mean Gaussian cells per map the split stands for: of the sub-batch it was first measured
                                       // on; after a re-measurement, of the recent sub-batches that asked for it (a registrar
                                       // that is fed two kinds of scenes in turn settles on their mean instead of measuring
                                       // again at every change)
    double recal_ref = 0.0;
    size_t calib_at = 0;               // ... and its number
    unsigned long long *stat_host = nullptr; // [depth][2], pinned: {Gaussian cells of the slot's maps, sub-batch + 1} written by a
                                             // one-workgroup kernel behind the build (no copy engine, no event: a device-to-host
                                             // copy per sub-batch cost the pipeline a quarter of its rate, measured)
    std::vector<long long> stat_seq;   // sub-batch whose counters slot k was asked for (-1: none / consumed)
    std::vector<unsigned> stat_maps;
    std::vector<double> recent_cells;  // mean cells per map of the last sub-batches seen (at most `depth`)
    // host clouds (ndtgpu_register_batch_host): per slot a device staging area for the scans of a sub-batch, one for the
    // poses / results of a call, a copy stream
    std::vector<void *> hstage;
    std::vector<size_t> hstage_bytes;
    void *hio = nullptr;
    size_t hio_bytes = 0;
    hipStream_t hcopy = nullptr;
    bool profiling = false;
    struct ProfMark { hipEvent_t e[4]; long long seq; };   // build start / end, matcher start / end (events), or the queue's stamps of `seq`
    std::vector<ProfMark> marks;
};

static int device_cus()
{
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
        n_cu = 256;
    return n_cu;
}

ndtgpu_status ndtgpu_registrar_destroy(ndtgpu_registrar *r)
{
    if (!r) return NDTGPU_OK;
    for (hipStream_t st : r->streams)
        if (st) (void)hipStreamSynchronize(st);
    for (auto &m : r->marks)
        for (hipEvent_t e : m.e) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : r->built) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : r->done) if (e) (void)hipEventDestroy(e);
    if (r->in_ev) (void)hipEventDestroy(r->in_ev);
    if (r->hcopy) { (void)hipStreamSynchronize(r->hcopy); (void)hipStreamDestroy(r->hcopy); }
    if (r->bst) (void)hipStreamSynchronize(r->bst);
    if (r->bst2) (void)hipStreamSynchronize(r->bst2);
    if (r->pst) { (void)hipStreamSynchronize(r->pst); (void)hipStreamDestroy(r->pst); }
    if (r->hst) { (void)hipStreamSynchronize(r->hst); (void)hipStreamDestroy(r->hst); }
    if (r->mst) { (void)hipStreamSynchronize(r->mst); (void)hipStreamDestroy(r->mst); }
    if (r->bst) (void)hipStreamDestroy(r->bst);
    if (r->bst2) (void)hipStreamDestroy(r->bst2);
    for (hipEvent_t e : r->pub_ev) if (e) (void)hipEventDestroy(e);
    if (r->stat_host) (void)hipHostFree(r->stat_host);
    for (auto &m : r->masked) if (m.second) { (void)hipStreamSynchronize(m.second); (void)hipStreamDestroy(m.second); }
    for (hipEvent_t e : r->probe_ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : r->trace_ev) if (e) (void)hipEventDestroy(e);
    if (r->queue) (void)hipFree(r->queue);
    for (void *q : r->hstage) if (q) (void)hipFree(q);
    if (r->hio) (void)hipFree(r->hio);
    for (ndtgpu_mapset *s : r->sets) (void)ndtgpu_mapset_destroy(s);
    for (hipStream_t st : r->streams)
        if (st) (void)hipStreamDestroy(st);
    if (r->iota) (void)hipFree(r->iota);
    delete r;
    return NDTGPU_OK;
}

void ndtgpu_default_registrar_params(ndtgpu_registrar_params *p)
{
    if (!p) return;
    p->pairs_per_batch = 1024;
    p->depth = 8;
    p->matcher_form = NDTGPU_MATCHER_AUTO;
    p->matcher_groups = 0;
    p->build_streams = 0;
    p->linger_us = 0;
    p->recalibrate_pct = 25;
    p->matcher_slots = 0;
}

// An experiment's environment variable overrides a field the caller LEFT AT ITS DEFAULT (0 / auto); what a caller sets wins.
static int env_int(const char *name, int fallback)
{
    const char *e = getenv(name);
    return e ? atoi(e) : fallback;
}

ndtgpu_status ndtgpu_registrar_create_ex(const ndtgpu_grid_params *grid, const ndtgpu_registrar_params *params, ndtgpu_registrar **out)
{
    if (!grid || !params || !out) return fail(NDTGPU_ERR_INVALID, "registrar_create: null argument");
    ndtgpu_registrar_params P = *params;
    const size_t pairs_per_batch = P.pairs_per_batch;
    const int depth = P.depth;
    if (pairs_per_batch == 0 || pairs_per_batch > (1u << 30) || depth < 1 || depth > 16)
        return fail(NDTGPU_ERR_INVALID, "registrar_create: bad argument (pairs_per_batch >= 1, 1 <= depth <= 16)");
    if (P.matcher_form < NDTGPU_MATCHER_AUTO || P.matcher_form > NDTGPU_MATCHER_STREAM_FED || P.build_streams < 0 || P.build_streams > 2)
        return fail(NDTGPU_ERR_INVALID, "registrar_create: matcher_form must be 0..2, build_streams 0..2");
    if (!have_device()) return fail(NDTGPU_ERR_NO_DEVICE, "registrar_create: no HIP device");
    // experiments (tools/, A/B runs): only where the caller asked for the default
    if (P.matcher_form == NDTGPU_MATCHER_AUTO && getenv("NDTGPU_REG_STREAM"))
        P.matcher_form = env_int("NDTGPU_REG_STREAM", 1) ? NDTGPU_MATCHER_AUTO : NDTGPU_MATCHER_PER_BATCH;
    if (P.matcher_groups == 0) P.matcher_groups = (unsigned)std::max(0, env_int("NDTGPU_REG_GROUPS", 0));
    if (P.build_streams == 0) P.build_streams = std::min(2, std::max(0, env_int("NDTGPU_REG_BUILD_STREAMS", 0)));
    // An instance that has worked stays for `linger` when it runs dry.  Where the builds are the slower side (a split that gives the
    // matcher more than its share) a batch is complete before the next one is published; an instance that leaves then has to be
    // placed again -- 144 whole CUs among build workgroups that keep arriving -- and the pipeline falls into lockstep: measured on the
    // bench with 144 matcher CUs forced, 259 k registrations/s without linger, 560 k with 1 ms; at the measured split (128) linger
    // changes nothing (0 / 200 / 1000 us: 664 / 666 / 661 k).  Default 1 ms; ndtgpu_registrar_sync switches it off for what is
    // already submitted, so a waiting host does not pay for it.
    if (P.linger_us == 0) P.linger_us = (unsigned)std::max(0, env_int("NDTGPU_REG_LINGER_US", 1000));
    if (P.recalibrate_pct == 0) P.recalibrate_pct = 25;
    if (P.matcher_slots != 0 && P.matcher_slots != 2 && P.matcher_slots != 3)
        return fail(NDTGPU_ERR_INVALID, "registrar_create: matcher_slots must be 0 (auto), 2 or 3");
    if (P.matcher_slots == 0) { const int es = env_int("NDTGPU_REG_SLOTS", 0); if (es == 2 || es == 3) P.matcher_slots = es; }
    ndtgpu_registrar *r = new (std::nothrow) ndtgpu_registrar();
    if (!r) return fail(NDTGPU_ERR_ALLOC, "registrar_create: host alloc");
    r->per = pairs_per_batch;
    r->depth = depth;
    r->n_cu = device_cus();
    (void)hipGetDevice(&r->device);
    r->sets.assign(depth, nullptr);
    r->streams.assign(depth, nullptr);
    r->built.assign(depth, nullptr);
    r->done.assign(4 * (size_t)depth, nullptr);
    hipError_t e = hipSuccess;
    ndtgpu_status rc = NDTGPU_OK;
    for (int k = 0; k < depth && rc == NDTGPU_OK && e == hipSuccess; k++) {
        rc = ndtgpu_mapset_create(grid, 2 * pairs_per_batch, &r->sets[k]);
        if (rc != NDTGPU_OK) break;
        // Pipelined (depth > 1), a matcher launch keeps to half of the CUs: its persistent workgroups hold a CU each, whole, until
        // their registrations are done, and with all CUs taken the next sub-batch's builds would wait for the launch's first
        // exits (measured, 1024 pairs per sub-batch: 470 k registrations/s with 256 workgroups, 488 k with 128..160)
        if (depth > 1) r->sets[k]->match_groups = P.matcher_groups ? P.matcher_groups : (unsigned)(r->n_cu / 2);
        e = hipStreamCreateWithFlags(&r->streams[k], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&r->built[k], hipEventDisableTiming);
    }
    for (size_t k = 0; k < r->done.size() && rc == NDTGPU_OK && e == hipSuccess; k++)
        e = hipEventCreateWithFlags(&r->done[k], hipEventDisableTiming);
    // The stream-fed matcher: pipelined registrars of small maps.  It needs a hardware queue of its own for the matcher stream:
    // the registrar orders map-set reuse with device-side wait kernels that only end when the running instance makes progress,
    // so an instance launch must never sit in a queue behind such a kernel.  What keeps the streams apart is a priority of
    // their own each -- a device that offers one priority level only (lo == hi) keeps the form with one matcher launch per
    // sub-batch (ordered by events), as does NDTGPU_REG_PRIO=0.
    if (rc == NDTGPU_OK && e == hipSuccess) {
        int lo = 0, hi = 0;
        const bool prio_ok = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi && env_int("NDTGPU_REG_PRIO", 1) != 0;
        const bool can = depth > 1 && (unsigned)depth <= ndt_stream_ring() && r->sets[0]->v.grid.max_cells < 16384u && prio_ok;
        if (P.matcher_form == NDTGPU_MATCHER_STREAM_FED && !can) {
            ndtgpu_registrar_destroy(r);
            return fail(NDTGPU_ERR_INVALID, "registrar_create: the stream-fed matcher needs 2 <= depth <= 8, max_cells < 16384 and a device "
                                            "with more than one stream priority");
        }
        if (P.matcher_form != NDTGPU_MATCHER_PER_BATCH && can) {
            r->stream_groups = P.matcher_groups;            // 0: measured on the first sub-batch
            r->stream_slots = P.matcher_slots ? P.matcher_slots : 2;   // (auto: decided with the split, by the cells per map)
            e = hipMalloc(&r->queue, ndt_stream_queue_bytes());
            if (e == hipSuccess) e = hipMemset(r->queue, 0, ndt_stream_queue_bytes());
            const unsigned ring_linger[2] = {(unsigned)depth, 100u * P.linger_us};   // 100 MHz ticks (measured: no gain from 300 / 1000 us; default 0)
            if (e == hipSuccess) e = hipMemcpy((char *)r->queue + ndt_stream_ring_offset(), ring_linger, sizeof ring_linger, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->bst, hipStreamNonBlocking);
            // Two build streams that take the sub-batches in turn (build_streams = 1: one): a build launch is one
            // workgroup per map and every map costs about the same, so on F free CUs it takes ceil(maps / 4 F) whole rounds
            // (measured: 1.47 ms beside a matcher instance on 128 CUs, 1.85 ms beside one on 129); with the next launch's
            // workgroups filling the last, nearly empty round the build side runs at its average rate whatever F is.
            // Publishes stay in order on a stream of their own.
            if (P.build_streams == 0) P.build_streams = depth >= 3 ? 2 : 1;
            if (depth < 3) P.build_streams = 1;
            if (e == hipSuccess && P.build_streams == 2) e = hipStreamCreateWithFlags(&r->bst2, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&r->pst, hipStreamNonBlocking, lo);
            // (a priority of its own: the runtime then never maps the two streams onto one hardware queue, where the build of
            //  batch k + 1 would sit behind the running matcher instance)
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&r->mst, hipStreamNonBlocking, hi);
            r->mst_prio = hi;
            if (e == hipSuccess) (void)hipStreamGetPriority(r->mst, &r->mst_prio);
            r->pub_ev.assign(depth, nullptr);
            for (int k = 0; k < depth && e == hipSuccess; k++) e = hipEventCreateWithFlags(&r->pub_ev[k], hipEventDisableTiming);
            // the side channel of the map statistics
            if (e == hipSuccess && P.matcher_groups == 0 && P.recalibrate_pct > 0) {
                e = hipHostMalloc((void **)&r->stat_host, (size_t)depth * 2 * sizeof(unsigned long long), hipHostMallocDefault);
                if (e == hipSuccess) memset(r->stat_host, 0, (size_t)depth * 2 * sizeof(unsigned long long));
                r->stat_seq.assign(depth, -1);
                r->stat_maps.assign(depth, 0u);
            }
        } else {
            P.build_streams = 0;
        }
        P.matcher_form = r->queue ? NDTGPU_MATCHER_STREAM_FED : NDTGPU_MATCHER_PER_BATCH;
    }
    r->prm = P;
    if (rc == NDTGPU_OK && e == hipSuccess) e = hipEventCreateWithFlags(&r->in_ev, hipEventDisableTiming);
    if (rc == NDTGPU_OK && e == hipSuccess) e = hipMalloc((void **)&r->iota, (2 * pairs_per_batch + 4) * sizeof(uint32_t));
    if (rc == NDTGPU_OK && e == hipSuccess) {
        std::vector<uint32_t> h(2 * pairs_per_batch);
        for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)i;
        e = hipMemcpy(r->iota, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    }
    if (rc != NDTGPU_OK || e != hipSuccess) {
        const std::string why = rc != NDTGPU_OK ? g_err : std::string("registrar_create: ") + hipGetErrorString(e);
        ndtgpu_registrar_destroy(r);
        return fail(rc != NDTGPU_OK ? rc : NDTGPU_ERR_HIP, why.c_str());
    }
    *out = r;
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_registrar_create(const ndtgpu_grid_params *grid, size_t pairs_per_batch, int depth, ndtgpu_registrar **out)
{
    ndtgpu_registrar_params p;
    ndtgpu_default_registrar_params(&p);
    p.pairs_per_batch = pairs_per_batch;
    p.depth = depth;
    return ndtgpu_registrar_create_ex(grid, &p, out);
}

ndtgpu_status ndtgpu_registrar_get_info(const ndtgpu_registrar *r, ndtgpu_registrar_info *info)
{
    if (!r || !info) return fail(NDTGPU_ERR_INVALID, "registrar_get_info: null argument");
    info->matcher_form = r->prm.matcher_form;
    info->matcher_groups = r->queue ? r->stream_groups : r->sets[0]->match_groups;
    info->build_streams = r->prm.build_streams;
    info->calibrations = r->calibrations;
    info->submitted = (uint64_t)r->submitted;
    info->cells_per_map = r->calib_cells;
    info->matcher_slots = r->queue ? r->stream_slots : 2;
    info->resident_groups = 0;
    if (r->queue) {                                         // (a 4-byte read on the null stream; the registrar's streams do not block it)
        unsigned live = 0u;
        HIP_TRY(hipMemcpy(&live, (const char *)r->queue + ndt_stream_live_offset(), sizeof live, hipMemcpyDeviceToHost));
        info->resident_groups = (int32_t)live;
    }
    return NDTGPU_OK;
}

// test aid: raises the stream-fed matcher's abort word, as a workgroup that found no work for ~30 s would
ndtgpu_status ndtgpu_registrar_inject_abort(ndtgpu_registrar *r)
{
    if (!r) return fail(NDTGPU_ERR_INVALID, "registrar_inject_abort: null");
    if (!r->queue) return fail(NDTGPU_ERR_INVALID, "registrar_inject_abort: not the stream-fed form");
    const unsigned one = 1u;
    HIP_TRY(hipMemcpy((char *)r->queue + ndt_stream_abort_offset(), &one, sizeof one, hipMemcpyHostToDevice));
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_registrar_mapset(ndtgpu_registrar *r, int slot, ndtgpu_mapset **set)
{
    if (!r || !set || slot < 0 || slot >= r->depth) return fail(NDTGPU_ERR_INVALID, "registrar_mapset: bad argument");
    *set = r->sets[slot];
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_registrar_profiling(ndtgpu_registrar *r, int on)
{
    if (!r) return fail(NDTGPU_ERR_INVALID, "registrar_profiling: null");
    r->profiling = on != 0;
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_registrar_kernel_ms(ndtgpu_registrar *r, float mean_ms[2], int32_t *launches)
{
    if (!r || !mean_ms || !launches) return fail(NDTGPU_ERR_INVALID, "registrar_kernel_ms: bad argument");
    const size_t n = r->marks.size();
    double sum[2] = {0.0, 0.0};
    size_t cnt[2] = {0, 0};
    if (r->queue && n) {                    // the stamps are complete once the matcher side is
        ndtgpu_status rc = ndtgpu_registrar_sync(r);
        if (rc != NDTGPU_OK) return rc;
    }
    for (size_t k = 0; k < n; k++) {
        ndtgpu_registrar::ProfMark &m = r->marks[k];
        HIP_TRY(hipEventSynchronize(m.e[m.seq >= 0 ? 1 : 3]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, m.e[0], m.e[1]));
        sum[0] += ms; cnt[0]++;
        if (m.seq < 0) {
            HIP_TRY(hipEventElapsedTime(&ms, m.e[2], m.e[3]));
            sum[1] += ms; cnt[1]++;
        } else if ((size_t)m.seq + ndt_stream_stamps() > r->submitted) {
            // stream-fed form: what the queue saw of the sub-batch -- published (its maps built) until its last registration finished
            unsigned long long st[2] = {0, 0};
            HIP_TRY(ndt_stream_read_stamps(r->queue, (unsigned)m.seq, st));
            if (st[1] > st[0]) { sum[1] += (double)(st[1] - st[0]) * 1e-5; cnt[1]++; }
        }
    }
    for (auto &m : r->marks)
        for (hipEvent_t e : m.e) if (e) (void)hipEventDestroy(e);
    r->marks.clear();
    *launches = (int32_t)n;
    mean_ms[0] = cnt[0] ? (float)(sum[0] / (double)cnt[0]) : 0.f;
    mean_ms[1] = cnt[1] ? (float)(sum[1] / (double)cnt[1]) : 0.f;
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_register_batch_device(ndtgpu_registrar *r, const void *targets_dev, const void *sources_dev, size_t n_points,
                                           size_t stride_bytes, size_t map_stride_bytes, double range_limit,
                                           const ndtgpu_cell_params *cell, double *T16_dev, size_t n_pairs,
                                           const ndtgpu_match_params *prm, ndtgpu_match_result *results_dev, ndtgpu_stream stream,
                                           uint64_t *ticket)
{
    if (ticket) *ticket = r ? (uint64_t)r->submitted : 0;
    if (!r || (n_pairs && (!T16_dev || !results_dev || (n_points && (!targets_dev || !sources_dev)))) || stride_bytes < 12 ||
        (stride_bytes & 3) || n_points > 0xFFFFFFFFull)
        return fail(NDTGPU_ERR_INVALID, "register_batch_device: bad argument");
    if (n_pairs == 0) return NDTGPU_OK;
    NdtMatchParamsDev pdev = to_dev(prm);
    pdev.fusion_flags = 0;
    if (pdev.n_neighbours < 0 || pdev.n_neighbours > 3 || (pdev.dof_mask & 0x3f) == 0)
        return fail(NDTGPU_ERR_INVALID, "match: n_neighbours must be 0..3 and dof_mask non-empty");
    HIP_TRY(hipEventRecord(r->in_ev, (hipStream_t)stream));
    auto new_mark = [&](long long seq, ndtgpu_registrar::ProfMark **out_mark) -> ndtgpu_status {
        ndtgpu_registrar::ProfMark m{};
        m.seq = seq;
        for (int k = 0; k < (seq >= 0 ? 2 : 4); k++) HIP_TRY(hipEventCreate(&m.e[k]));
        r->marks.push_back(m);
        *out_mark = &r->marks.back();
        return NDTGPU_OK;
    };
    // one build launch when the sources follow the targets in memory, else two
    auto build_pairs = [&](ndtgpu_mapset *set, size_t off, size_t p, hipStream_t st) -> ndtgpu_status {
        const char *tg = (const char *)targets_dev + off * map_stride_bytes, *sc = (const char *)sources_dev + off * map_stride_bytes;
        if (sc == tg + p * map_stride_bytes)
            return ndtgpu_mapset_build(set, 0, 2 * p, tg, n_points, stride_bytes, map_stride_bytes, range_limit, nullptr, cell, st);
        ndtgpu_status rc = ndtgpu_mapset_build(set, 0, p, tg, n_points, stride_bytes, map_stride_bytes, range_limit, nullptr, cell, st);
        if (rc == NDTGPU_OK)
            rc = ndtgpu_mapset_build(set, p, p, sc, n_points, stride_bytes, map_stride_bytes, range_limit, nullptr, cell, st);
        return rc;
    };
    if (r->queue) {
        // ---- stream-fed form: builds on one stream, batches published to the running matcher instance -------------------
        auto drain = [&]() -> ndtgpu_status {
            HIP_TRY(hipStreamSynchronize(r->bst));
            if (r->bst2) HIP_TRY(hipStreamSynchronize(r->bst2));
            HIP_TRY(hipStreamSynchronize(r->pst));
            HIP_TRY(hipStreamSynchronize(r->mst));
            if (r->hst) HIP_TRY(hipStreamSynchronize(r->hst));
            return NDTGPU_OK;
        };
        if (r->stream_nn != pdev.n_neighbours) {                 // (an instance is compiled for one neighbourhood size)
            if (r->stream_nn >= 0) { ndtgpu_status drc = drain(); if (drc != NDTGPU_OK) return drc; }
            r->stream_nn = pdev.n_neighbours;
        }
        for (size_t off = 0; off < n_pairs; off += r->per) {
            const size_t p = std::min(r->per, n_pairs - off);
            const size_t j = r->submitted;
            const int slot = (int)(j % (size_t)r->depth);
            ndtgpu_mapset *set = r->sets[slot];
            hipStream_t st = (r->bst2 && (j & 1u)) ? r->bst2 : r->bst;
            // ---- have the maps changed?  The counters of earlier builds that have arrived on the host say how many Gaussian
            // cells a map holds now; when the mean over the last sub-batches has left the figure the split was measured at by
            // more than recalibrate_pct, the pipeline is drained once and this sub-batch measures the split again (a
            // registrar that moves from halls to clutter would otherwise keep 128 matcher CUs where 200 are right).
            if (r->stat_host && r->stream_groups != 0u) {
                // (back-pressure: the host runs at most `depth` sub-batches ahead of the builds -- without it a caller that
                //  never waits would have submitted everything before the first counters arrive)
                if (j >= (size_t)r->depth) HIP_TRY(hipEventSynchronize(r->built[slot]));
                for (int k = 0; k < r->depth; k++) {
                    volatile unsigned long long *sh = r->stat_host + 2 * k;
                    if (r->stat_seq[k] < 0 || sh[1] != (unsigned long long)r->stat_seq[k] + 1ull) continue;
                    std::atomic_thread_fence(std::memory_order_acquire);
                    if (r->stat_maps[k]) {
                        if (r->recent_cells.size() >= (size_t)r->depth) r->recent_cells.erase(r->recent_cells.begin());
                        r->recent_cells.push_back((double)sh[0] / (double)r->stat_maps[k]);
                    }
                    r->stat_seq[k] = -1;
                }
                if (r->recent_cells.size() >= (size_t)std::min(r->depth, 4) && r->calib_cells > 0 && j >= r->calib_at + 2 * (size_t)r->depth) {
                    double mean = 0;
                    for (double v : r->recent_cells) mean += v;
                    mean /= (double)r->recent_cells.size();
                    if (std::fabs(mean / r->calib_cells - 1.0) * 100.0 > (double)r->prm.recalibrate_pct) {
                        if (getenv("NDTGPU_REG_VERBOSE"))
                            fprintf(stderr, "ndtgpu registrar: %.0f cells per map where the split was measured at %.0f: measuring again\n", mean, r->calib_cells);
                        ndtgpu_status drc = drain();
                        if (drc != NDTGPU_OK) return drc;
                        r->stream_groups = 0u;
                        r->recal_ref = mean;
                    }
                }
            }
            HIP_TRY(hipStreamWaitEvent(st, r->in_ev, 0));
            if (r->stream_groups == 0u) {
                // ---- a sub-batch that measures the split of the chip (the first of a registrar's life; later ones after a
                // drain, see above): its maps are built and its pairs registered with nothing else on the device (the whole
                // chip each; the same bits), the host reads the kernels' own clocks -- CU-time of the builds B and of the
                // registrations M -- and a matcher instance gets n_cu M / (M + B) CUs from then on: 128 of 256 on the bench's
                // halls, 200 on a cluttered scene whose maps hold five times the cells.  Costs one synchronisation.
                ndtgpu_registrar::ProfMark *mk0 = nullptr;
                if (r->profiling) { ndtgpu_status mrc = new_mark(-1, &mk0); if (mrc != NDTGPU_OK) return mrc; HIP_TRY(hipEventRecord(mk0->e[0], st)); }
                ndtgpu_status rc0 = build_pairs(set, off, p, st);
                if (rc0 != NDTGPU_OK) return rc0;
                if (mk0) { HIP_TRY(hipEventRecord(mk0->e[1], st)); HIP_TRY(hipEventRecord(mk0->e[2], st)); }
                const unsigned saved_groups = set->match_groups;
                set->match_groups = 0;                            // (the whole chip)
                rc0 = ndtgpu_match_batch_device(set, r->iota, set, r->iota + p, T16_dev + off * 16, p, prm, results_dev + off, st);
                set->match_groups = saved_groups;
                if (rc0 != NDTGPU_OK) return rc0;
                if (mk0) HIP_TRY(hipEventRecord(mk0->e[3], st));
                hipError_t se = ndt_stream_skip(r->queue, (unsigned)j, st);
                if (se != hipSuccess) return fail(NDTGPU_ERR_HIP, "registrar: calibration", se);
                HIP_TRY(hipEventRecord(r->built[slot], st));
                HIP_TRY(hipStreamSynchronize(st));
                std::vector<NdtMapCounters> ctr(2 * p);
                std::vector<ndtgpu_match_result> res(p);
                HIP_TRY(hipMemcpy(ctr.data(), set->v.counters, 2 * p * sizeof(NdtMapCounters), hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(res.data(), results_dev + off, p * sizeof(ndtgpu_match_result), hipMemcpyDeviceToHost));
                double B = 0, M = 0, cells = 0;
                for (const NdtMapCounters &c : ctr) {
                    B += (double)c.cyc[0] + (double)c.cyc[1] + (double)c.cyc[2] + (double)c.cyc[3];
                    cells += (double)c.n_cells;
                }
                B /= 4.0;                                         // a build workgroup shares its CU with three others
                // (a clock sum outside any plausible range -- seen once in six runs on the cluttered scene: 2^63 in one result -- is
                //  left out: the split is a heuristic, one registration does not move it; NDTGPU_REG_VERBOSE reports it)
                size_t m_bad = 0;
                for (const ndtgpu_match_result &q : res) {
                    const bool sane = q.cycles_eval >= 0 && q.cycles_eval < (1ll << 44) && q.cycles_solver >= 0 && q.cycles_solver < (1ll << 44);
                    if (sane) M += (double)q.cycles_eval + (double)q.cycles_solver / 8.0;
                    else {
                        if (getenv("NDTGPU_REG_VERBOSE") && m_bad < 4)
                            fprintf(stderr, "ndtgpu registrar: calibration result %zu: cycles_eval %lld cycles_solver %lld iterations %d exit %d\n",
                                    (size_t)(&q - res.data()), (long long)q.cycles_eval, (long long)q.cycles_solver, (int)q.iterations, (int)q.exit_code);
                        m_bad++;
                    }
                }
                if (m_bad < res.size()) M *= (double)res.size() / (double)(res.size() - m_bad);
                const double M_raw = M;                           // CU-clocks of the sub-batch's registrations
                M *= 1.125;                                       // (measured optimum on the bench scene: 144 of 256 CUs where the raw clocks say 138)
                const int n_cu = r->n_cu;
                double share = (M + B) > 0 ? M / (M + B) : 0.5;
                share = std::min(0.9, std::max(0.25, share));
                r->stream_groups = std::max(8u, ((unsigned)(share * n_cu + 4.0) / 8u) * 8u);
                // ... refined by what the build side really does with the CUs it is left: a build launch is one workgroup per
                // map, four to a CU, so on F CUs it takes ceil(maps / 4F) ROUNDS of the mean workgroup time -- 2048 maps take
                // four rounds on 128 CUs and five on 120, 112 or 104.  Of the splits around the proportional one, take the one
                // whose slower side is fastest (the bench halls after the round's matcher savings: 136 by proportion, 573 k
                // registrations/s; 128 by this rule, 624 k; 120: 585 k, 144: 544 k).
                if (B > 0 && M > 0) {
                    const double wg = 4.0 * B / (2.0 * (double)p);            // mean clocks of a build workgroup
                    double best_t = 0;
                    unsigned best_g = r->stream_groups;
                    for (int g8 = (int)r->stream_groups - 32; g8 <= (int)r->stream_groups + 32; g8 += 8) {
                        if (g8 < 16 || g8 > n_cu - 16) continue;
                        const double slots = 4.0 * (n_cu - g8);
                        const double t = std::max(M / g8, std::ceil(2.0 * (double)p / slots) * wg);
                        const bool closer = std::abs(g8 - (int)r->stream_groups) < std::abs((int)best_g - (int)r->stream_groups);
                        if (best_t == 0 || t < 0.99 * best_t || (t <= 1.01 * best_t && closer && t <= best_t)) { best_t = t; best_g = (unsigned)g8; }
                    }
                    r->stream_groups = best_g;
                }
                // ... and, since 0.6.5, MEASURED on the build side (round 6).  The clocks above are those of workgroups that had the
                // whole chip: beside a matcher instance the builds of the bench halls run 20 % faster than that (half as many
                // workgroups pull on the HBM), those of a cluttered scene 30 % slower (four workgroups to a CU where the lone launch
                // had three), and a quarter more or less decides between two splits (halls 120 instead of 128 CUs on one box in
                // three: -5 %; clutter 184 instead of 160: 103 against 127 k registrations/s).  A build launch takes whole rounds,
                // so the only splits worth having are those that leave the builds just enough CUs for k rounds, k = 1, 2, ...:
                // the sub-batch is built again on a stream that owns exactly those CUs (hipExtStreamCreateWithCUMask: mask bits
                // are dealt to the XCDs in turn, like the CUs a matcher instance leaves), timed with events, and the split whose
                // slower side -- that time, or the registrations' CU-clocks over the matcher's CUs -- is fastest wins.  A few
                // build launches, once per measurement; NDTGPU_REG_PROBE=0 keeps the model above.
                if (B > 0 && M_raw > 0 && env_int("NDTGPU_REG_PROBE", 1) != 0) {
                    int khz = 0;
                    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, r->device) != hipSuccess || khz <= 0) khz = 2400000;
                    const double clk_per_ms = 0.85 * (double)khz;              // (the clock under these kernels: 1.93-2.1 of 2.4 GHz)
                    const size_t maps = 2 * p;
                    struct Cand { unsigned g; double build_ms, t_ms; };
                    std::vector<Cand> cand;
                    auto groups_of = [&](unsigned k) -> unsigned {             // the most matcher CUs that leave the builds k rounds
                        const unsigned F = (unsigned)((maps + 4u * k - 1u) / (4u * k));
                        const unsigned F8 = (F + 7u) / 8u * 8u;
                        return ((unsigned)n_cu > F8 + 15u) ? (unsigned)n_cu - F8 : 0u;
                    };
                    auto probe = [&](unsigned g, double *ms) -> ndtgpu_status {
                        hipStream_t ms_st = nullptr;
                        ndtgpu_status prc = r->masked_stream((unsigned)n_cu - g, &ms_st);
                        if (prc != NDTGPU_OK) return prc;
                        if (!r->probe_ev[0]) { HIP_TRY(hipEventCreate(&r->probe_ev[0])); HIP_TRY(hipEventCreate(&r->probe_ev[1])); }
                        float best = 0.f;
                        for (int rep = 0; rep < 2; rep++) {                    // (the first launch on a new stream pays for the stream)
                            HIP_TRY(hipEventRecord(r->probe_ev[0], ms_st));
                            prc = build_pairs(set, off, p, ms_st);
                            if (prc != NDTGPU_OK) return prc;
                            HIP_TRY(hipEventRecord(r->probe_ev[1], ms_st));
                            HIP_TRY(hipEventSynchronize(r->probe_ev[1]));
                            float e = 0.f;
                            HIP_TRY(hipEventElapsedTime(&e, r->probe_ev[0], r->probe_ev[1]));
                            if (rep == 0 || e < best) best = e;
                        }
                        *ms = (double)best;
                        return NDTGPU_OK;
                    };
                    // the round count the model's split stands for, and its neighbours; further out while the edge keeps winning
                    unsigned k0 = 1;
                    for (unsigned k = 1; k <= 16; k++) { k0 = k; if (groups_of(k) >= r->stream_groups) break; }
                    bool ok = true;
                    auto have = [&](unsigned g) { for (const Cand &c : cand) if (c.g == g) return true; return false; };
                    auto add = [&](unsigned k) {
                        const unsigned g = k >= 1 && k <= 16 ? groups_of(k) : 0u;
                        if (!ok || g < 16u || have(g)) return;
                        Cand c{g, 0.0, 0.0};
                        if (probe(g, &c.build_ms) != NDTGPU_OK) { ok = false; return; }
                        c.t_ms = std::max(c.build_ms, M_raw / (double)g / clk_per_ms);
                        cand.push_back(c);
                    };
                    auto best_of = [&]() { size_t b = 0; for (size_t i = 1; i < cand.size(); i++) if (cand[i].t_ms < cand[b].t_ms) b = i; return b; };
                    add(k0); add(k0 > 1 ? k0 - 1 : k0 + 2); add(k0 + 1);
                    for (int more = 0; ok && more < 3 && !cand.empty(); more++) {
                        const Cand &b = cand[best_of()];
                        unsigned kb = 0;
                        for (unsigned k = 1; k <= 16; k++) if (groups_of(k) == b.g) { kb = k; break; }
                        const size_t n_before = cand.size();
                        if (kb > 1 && !have(groups_of(kb - 1))) add(kb - 1);
                        if (kb && kb < 16 && !have(groups_of(kb + 1))) add(kb + 1);
                        if (cand.size() == n_before) break;
                    }
                    if (ok && !cand.empty()) {
                        const Cand &b = cand[best_of()];
                        if (getenv("NDTGPU_REG_VERBOSE"))
                            for (const Cand &c : cand)
                                fprintf(stderr, "ndtgpu registrar: %u matcher CUs: builds %.3f ms on the other %u, registrations %.3f ms -> %.3f ms per sub-batch%s\n",
                                        c.g, c.build_ms, (unsigned)n_cu - c.g, M_raw / (double)c.g / clk_per_ms, c.t_ms, c.g == b.g ? "  <-" : "");
                        r->stream_groups = b.g;
                    } else if (!ok) {
                        (void)hipGetLastError();                               // (no masked streams on this device / runtime: the model's split stands)
                        g_err.clear();
                    }
                    // (the probe streams go at once: each is a hardware queue of its own and of no use until the next measurement;
                    //  creating them is most of what a measurement costs, ~50 ms with three or four candidates)
                    for (auto &m : r->masked) if (m.second) { (void)hipStreamSynchronize(m.second); (void)hipStreamDestroy(m.second); }
                    r->masked.clear();
                    HIP_TRY(hipEventRecord(r->built[slot], st));               // (the maps were rebuilt: same contents)
                }
                r->calibrations++;
                // three registrations per workgroup (hit lists of 640 entries per share) where the maps are small -- up to 448 cells: the
                // one-lane solver steps are a third of a registration there --, two with lists of 1024 entries otherwise (measured
                // on the cluttered scene: three are slower)
                if (r->prm.matcher_slots == 0) r->stream_slots = (cells / (double)(2 * p) <= 448.0) ? 3 : 2;
                r->calib_cells = r->recal_ref > 0 ? r->recal_ref : cells / (double)(2 * p);
                r->recal_ref = 0.0;
                r->calib_at = j;
                r->recent_cells.clear();
                for (long long &q : r->stat_seq) q = -1;
                if (getenv("NDTGPU_REG_VERBOSE")) fprintf(stderr, "ndtgpu registrar: build %.3g, registrations %.3g CU-clocks per sub-batch, %.0f cells per map -> matcher instances of %u workgroups\n", B, M, r->calib_cells, r->stream_groups);
                r->submitted++;
                if (ticket) *ticket = (uint64_t)r->submitted;
                continue;
            }
            // This map set was last used by sub-batch j - depth: its registrations must be complete before it is rebuilt.
            // (Depth: a batch is complete 3-5 ms after its publication; with 8 map sets the builds never wait for that, 4 cost
            //  ~5 % on the bench -- include/ndtgpu.h.)
            if (j >= (size_t)r->depth) {
                hipError_t we = ndt_stream_wait(r->queue, (unsigned)r->depth, (unsigned)(j - (size_t)r->depth), st);
                if (we != hipSuccess) return fail(NDTGPU_ERR_HIP, "registrar: wait launch", we);
            }
            ndtgpu_registrar::ProfMark *mk = nullptr;
            if (r->profiling) { ndtgpu_status mrc = new_mark((long long)j, &mk); if (mrc != NDTGPU_OK) return mrc; HIP_TRY(hipEventRecord(mk->e[0], st)); }
            const bool tracing = getenv("NDTGPU_REG_TRACE") != nullptr;
            if (tracing) {
                if (r->trace_ev.empty()) { r->trace_ev.assign(128, nullptr); for (hipEvent_t &e : r->trace_ev) HIP_TRY(hipEventCreate(&e)); }
                HIP_TRY(hipEventRecord(r->trace_ev[2 * (j % 64)], st));
                if (r->trace_first == (size_t)-1) r->trace_first = j;
            }
            ndtgpu_status rc = build_pairs(set, off, p, st);
            if (rc != NDTGPU_OK) return rc;
            if (tracing) HIP_TRY(hipEventRecord(r->trace_ev[2 * (j % 64) + 1], st));
            if (mk) HIP_TRY(hipEventRecord(mk->e[1], st));
            if (r->stat_host && r->stat_seq[slot] < 0 && j % 3u == 0u) {
                // (how many Gaussian cells the maps of this build hold: one small kernel behind it writes the sum to the host.
                //  Every third sub-batch: the launch costs the build stream a few microseconds -- 2 % of the bench's rate when
                //  every sub-batch had one -- and an odd period does not lock onto callers that alternate between two scenes)
                hipLaunchKernelGGL(ndt_reg_stats_kernel, dim3(1), dim3(256), 0, st, set->v.counters, (unsigned)(2 * p), (unsigned long long)j,
                                   r->stat_host + 2 * slot);
                HIP_TRY(hipGetLastError());
                r->stat_seq[slot] = (long long)j;
                r->stat_maps[slot] = (unsigned)(2 * p);
            }
            HIP_TRY(hipEventRecord(r->built[slot], st));
            HIP_TRY(hipStreamWaitEvent(r->pst, r->built[slot], 0));
            hipError_t pe = ndt_stream_publish(r->queue, set->v, T16_dev + off * 16, reinterpret_cast<NdtMatchResultDev *>(results_dev + off),
                                               pdev, (unsigned)p, (unsigned)j, r->pst);
            if (pe != hipSuccess) return fail(NDTGPU_ERR_HIP, "registrar: publish", pe);
            HIP_TRY(hipEventRecord(r->pub_ev[slot], r->pst));
            // every published batch is followed by an instance launch: it starts when the running instance has ended (and
            // then serves this batch and whatever is published while it runs), or finds the batch taken and leaves
            HIP_TRY(hipStreamWaitEvent(r->mst, r->pub_ev[slot], 0));
            pe = ndt_launch_match_stream(r->queue, pdev.n_neighbours, r->stream_slots, r->stream_groups, r->mst);
            if (pe != hipSuccess) return fail(NDTGPU_ERR_HIP, "registrar: matcher launch", pe);
            r->submitted++;
            if (ticket) *ticket = (uint64_t)r->submitted;
        }
        return NDTGPU_OK;
    }
    for (size_t off = 0; off < n_pairs; off += r->per) {
        const size_t p = std::min(r->per, n_pairs - off);
        const int slot = (int)(r->submitted % (size_t)r->depth);
        hipStream_t st = r->streams[slot];
        ndtgpu_mapset *set = r->sets[slot];
        HIP_TRY(hipStreamWaitEvent(st, r->in_ev, 0));
        if (r->last_built >= 0 && r->last_built != slot) HIP_TRY(hipStreamWaitEvent(st, r->built[r->last_built], 0));
        ndtgpu_registrar::ProfMark *mk = nullptr;
        if (r->profiling) { ndtgpu_status mrc = new_mark(-1, &mk); if (mrc != NDTGPU_OK) return mrc; HIP_TRY(hipEventRecord(mk->e[0], st)); }
        ndtgpu_status rc = build_pairs(set, off, p, st);
        if (rc != NDTGPU_OK) return rc;
        if (mk) { HIP_TRY(hipEventRecord(mk->e[1], st)); HIP_TRY(hipEventRecord(mk->e[2], st)); }
        HIP_TRY(hipEventRecord(r->built[slot], st));
        r->last_built = slot;
        // `built` releases the next sub-batch's build AND, on this stream, this sub-batch's matcher.  The matcher's persistent
        // workgroups (one per CU, all registers and LDS of it) must not be placed first: the build would then only get the CUs
        // the matcher leaves, its end -- which releases the build after it -- moves out, and the pipeline loses a tenth of its
        // rate (measured: 430 k against 470 k registrations/s).  Two 4-byte fills keep this stream busy for the few
        // microseconds the next build's dispatch needs to get ahead (NDTGPU_REG_GAP: their number).
        if (r->depth > 1) {
            const int n_gap = env_int("NDTGPU_REG_GAP", 2);
            for (int g = 0; g < n_gap; g++) HIP_TRY(hipMemsetAsync(r->iota + 2 * r->per, 0, 4, st));
        }
        rc = ndtgpu_match_batch_device(set, r->iota, set, r->iota + p, T16_dev + off * 16, p, prm, results_dev + off, st);
        if (rc != NDTGPU_OK) return rc;
        if (mk) HIP_TRY(hipEventRecord(mk->e[3], st));
        HIP_TRY(hipEventRecord(r->done[r->submitted % r->done.size()], st));
        r->submitted++;
        if (ticket) *ticket = (uint64_t)r->submitted;      // "every sub-batch before this count"
    }
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_registrar_wait_stream(ndtgpu_registrar *r, uint64_t ticket, ndtgpu_stream stream)
{
    if (!r || ticket > (uint64_t)r->submitted) return fail(NDTGPU_ERR_INVALID, "registrar_wait_stream: bad argument");
    const size_t end = ticket ? (size_t)ticket : r->submitted;
    if (r->queue) {
        // The wait is a device-side kernel that ends when the running matcher instance has made the batch complete: it must not
        // sit in the hardware queue the instance launches go through.  The runtime keeps streams of different priorities on
        // different queues; a stream of the matcher stream's priority (the highest the device offers) is refused.
        if (stream) {
            int prio = 0;
            if (hipStreamGetPriority((hipStream_t)stream, &prio) == hipSuccess && prio == r->mst_prio)
                return fail(NDTGPU_ERR_INVALID, "registrar_wait_stream: a stream of the highest priority may share the matcher's hardware queue; "
                                                "wait on a stream of default priority (or use ndtgpu_registrar_sync)");
        }
        // the last `depth` sub-batches before `end` (a sub-batch is only published once the one `depth` before it is complete)
        for (size_t j = end > (size_t)r->depth ? end - (size_t)r->depth : 0; j < end; j++) {
            hipError_t we = ndt_stream_wait(r->queue, (unsigned)r->depth, (unsigned)j, (hipStream_t)stream);
            if (we != hipSuccess) return fail(NDTGPU_ERR_HIP, "registrar: wait launch", we);
        }
        return NDTGPU_OK;
    }
    // the newest sub-batch before `end` on every internal stream (earlier ones precede it in stream order)
    for (size_t j = end > (size_t)r->depth ? end - (size_t)r->depth : 0; j < end; j++)
        HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, r->done[j % r->done.size()], 0));
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_registrar_sync(ndtgpu_registrar *r)
{
    if (!r) return fail(NDTGPU_ERR_INVALID, "registrar_sync: null");
    if (r->queue) {
        // Nothing more is coming before this call returns: once the last sub-batch has been published the CUs that were kept
        // for the builds are free, and a second instance on those takes its share of what is left to register.
        // (nothing more is coming before this call returns: instances do not linger behind the last published batch)
        if (r->submitted) {
            hipError_t fe = ndt_stream_final(r->queue, (unsigned)r->submitted, r->pst);
            if (fe != hipSuccess) return fail(NDTGPU_ERR_HIP, "registrar: final launch", fe);
        }
        if (r->submitted > r->helped && r->stream_groups && r->stream_nn >= 0 && !getenv("NDTGPU_REG_NO_HELPER")) {
            const int n_cu = r->n_cu;
            if ((unsigned)n_cu > r->stream_groups + 8u) {
                if (!r->hst) HIP_TRY(hipStreamCreateWithFlags(&r->hst, hipStreamNonBlocking));
                HIP_TRY(hipStreamWaitEvent(r->hst, r->pub_ev[(r->submitted - 1) % (size_t)r->depth], 0));
                hipError_t he = ndt_launch_match_stream(r->queue, r->stream_nn, r->stream_slots, (unsigned)n_cu - r->stream_groups, r->hst);
                if (he != hipSuccess) return fail(NDTGPU_ERR_HIP, "registrar: helper launch", he);
            }
            r->helped = r->submitted;
        }
        HIP_TRY(hipStreamSynchronize(r->bst));
        if (r->bst2) HIP_TRY(hipStreamSynchronize(r->bst2));
        HIP_TRY(hipStreamSynchronize(r->pst));
        HIP_TRY(hipStreamSynchronize(r->mst));
        if (r->hst) HIP_TRY(hipStreamSynchronize(r->hst));
        if (getenv("NDTGPU_REG_TRACE") && r->submitted) {
            // (experiments: when the last batches were published -- their maps built -- and when their last registration finished,
            //  in microseconds after the first of them)
            const size_t nb = std::min<size_t>(r->submitted, ndt_stream_stamps());
            unsigned long long t0 = 0;
            for (size_t k = r->submitted - nb; k < r->submitted; k++) {
                unsigned long long st[2] = {0, 0};
                HIP_TRY(ndt_stream_read_stamps(r->queue, (unsigned)k, st));
                if (!t0) t0 = st[0];
                float b0 = 0.f, b1 = 0.f;          // the build's start and end, against the end of the first listed build (~ its publication)
                const size_t kref = std::max(r->submitted - nb, r->trace_first);
                if (!r->trace_ev.empty() && k >= kref) {
                    (void)hipEventElapsedTime(&b0, r->trace_ev[2 * (kref % 64) + 1], r->trace_ev[2 * (k % 64)]);
                    (void)hipEventElapsedTime(&b1, r->trace_ev[2 * (kref % 64) + 1], r->trace_ev[2 * (k % 64) + 1]);
                    (void)hipGetLastError();
                }
                if (k == kref) t0 = st[0];
                fprintf(stderr, "[ndtgpu trace] batch %zu build %+.1f .. %+.1f us, published %+.1f us, done %+.1f us\n", k, 1e3 * b0, 1e3 * b1,
                        ((double)st[0] - (double)t0) * 0.01, ((double)st[1] - (double)t0) * 0.01);
            }
        }
        unsigned aborted = 0;
        HIP_TRY(hipMemcpy(&aborted, (char *)r->queue + ndt_stream_abort_offset(), sizeof aborted, hipMemcpyDeviceToHost));
        if (aborted) {
            // Reported ONCE: the registrations of the batches that were cut short carry exit_code -4 (every result starts as
            // "not run" when its batch is published), the queue is put back to "everything submitted is over", and the
            // registrar can be used again.
            HIP_TRY(ndt_stream_reset(r->queue, (unsigned)r->submitted, (unsigned)r->depth));
            return fail(NDTGPU_ERR_HIP, "registrar: the stream-fed matcher gave up (no work, or no progress behind a wait, for ~30 s); "
                                        "registrations that did not run report exit_code -4");
        }
        return NDTGPU_OK;
    }
    for (int k = 0; k < r->depth; k++) HIP_TRY(hipStreamSynchronize(r->streams[k]));
    for (int k = 0; k < r->depth && (size_t)k < r->submitted; k++) {
        int aborted = 0;
        ndtgpu_status rc = ndtgpu_match_aborted(r->sets[k], &aborted);
        if (rc != NDTGPU_OK) return rc;
        if (aborted) return fail(NDTGPU_ERR_HIP, "registrar: a matcher launch gave up (a wave found no work for ~1 s)");
    }
    return NDTGPU_OK;
}

// Host clouds in, host poses out: the reference's call sites hold pcl::PointCloud objects in host memory.  Sub-batch after
// sub-batch the scans travel to a device staging area of the slot they will be built in (one copy stream; the copies of
// sub-batch j + 1 run under the builds and registrations of sub-batch j), then the device entry takes over; poses and results
// come back with one copy each when everything is done.  Synchronous.
ndtgpu_status ndtgpu_register_batch_host(ndtgpu_registrar *r, const void *targets_host, const void *sources_host, size_t n_points,
                                         size_t stride_bytes, size_t map_stride_bytes, double range_limit,
                                         const ndtgpu_cell_params *cell, double *T16, size_t n_pairs, const ndtgpu_match_params *prm,
                                         ndtgpu_match_result *results)
{
    if (!r || (n_pairs && (!T16 || !results || (n_points && (!targets_host || !sources_host)))) || stride_bytes < 12 ||
        (stride_bytes & 3) || n_points > 0xFFFFFFFFull || (n_pairs > 1 && map_stride_bytes < n_points * stride_bytes))
        return fail(NDTGPU_ERR_INVALID, "register_batch_host: bad argument (clouds must not overlap: map_stride_bytes >= n_points * stride_bytes)");
    if (n_pairs == 0) return NDTGPU_OK;
    if (!r->hcopy) HIP_TRY(hipStreamCreateWithFlags(&r->hcopy, hipStreamNonBlocking));
    if (r->hstage.empty()) { r->hstage.assign(r->depth, nullptr); r->hstage_bytes.assign(r->depth, 0); }
    const size_t bT = n_pairs * 16 * sizeof(double), bR = n_pairs * sizeof(ndtgpu_match_result), offR = (bT + 255) & ~(size_t)255;
    if (r->hio_bytes < offR + bR) {
        HIP_TRY(hipStreamSynchronize(r->hcopy));
        if (r->hio) (void)hipFree(r->hio);
        r->hio = nullptr; r->hio_bytes = 0;
        HIP_TRY(hipMalloc(&r->hio, offR + bR));
        r->hio_bytes = offR + bR;
    }
    double *T_dev = (double *)r->hio;
    ndtgpu_match_result *R_dev = (ndtgpu_match_result *)((char *)r->hio + offR);
    HIP_TRY(hipMemcpyAsync(T_dev, T16, bT, hipMemcpyHostToDevice, r->hcopy));
    const size_t cloud_bytes = n_points * stride_bytes;
    for (size_t off = 0; off < n_pairs; off += r->per) {
        const size_t p = std::min(r->per, n_pairs - off);
        const int slot = (int)(r->submitted % (size_t)r->depth);
        const size_t half = (p - 1) * map_stride_bytes + cloud_bytes, half_al = p * map_stride_bytes;   // targets, then sources
        const size_t need = half_al + half;
        // the staging area of this slot is read by the build of the sub-batch that used it last: wait for that build
        if (r->submitted >= (size_t)r->depth) HIP_TRY(hipEventSynchronize(r->built[slot]));
        if (r->hstage_bytes[slot] < need) {
            if (r->hstage[slot]) (void)hipFree(r->hstage[slot]);
            r->hstage[slot] = nullptr; r->hstage_bytes[slot] = 0;
            HIP_TRY(hipMalloc(&r->hstage[slot], need));
            r->hstage_bytes[slot] = need;
        }
        char *tg = (char *)r->hstage[slot], *sc = tg + half_al;        // sources follow targets: ONE build launch per sub-batch
        if (n_points) {
            HIP_TRY(hipMemcpyAsync(tg, (const char *)targets_host + off * map_stride_bytes, half, hipMemcpyHostToDevice, r->hcopy));
            HIP_TRY(hipMemcpyAsync(sc, (const char *)sources_host + off * map_stride_bytes, half, hipMemcpyHostToDevice, r->hcopy));
        }
        // (p <= pairs_per_batch: ONE sub-batch, in this slot, behind these copies)
        ndtgpu_status rc = ndtgpu_register_batch_device(r, tg, sc, n_points, stride_bytes, map_stride_bytes, range_limit, cell,
                                                        T_dev + off * 16, p, prm, R_dev + off, (ndtgpu_stream)r->hcopy, nullptr);
        if (rc != NDTGPU_OK) return rc;
    }
    ndtgpu_status rc = ndtgpu_registrar_sync(r);
    if (rc != NDTGPU_OK) return rc;
    HIP_TRY(hipMemcpy(T16, T_dev, bT, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(results, R_dev, bR, hipMemcpyDeviceToHost));
    return NDTGPU_OK;
}

// host arrays -> staging -> persistent matcher -> host arrays; synchronous
static ndtgpu_status match_persistent_host(ndtgpu_mapset *ts, const uint32_t *tidx, ndtgpu_mapset *ss, const uint32_t *sidx,
                                           double *T16, size_t n_pairs, const NdtMatchParamsDev &p, const double *Q36,
                                           ndtgpu_match_result *results, hipStream_t st,
                                           const uint32_t *feat_off = nullptr, const double *feat_cells = nullptr)
{
    size_t bT = n_pairs * 16 * sizeof(double), bR = n_pairs * sizeof(ndtgpu_match_result), bI = n_pairs * sizeof(uint32_t);
    const size_t bQ_ = Q36 ? n_pairs * 36 * sizeof(double) : 0;
    const size_t bFo = feat_off ? (n_pairs + 1) * sizeof(uint32_t) : 0, bFc = feat_off ? (size_t)feat_off[n_pairs] * 18 * sizeof(double) : 0;
    size_t off_R = (bT + 255) & ~(size_t)255, off_ti = (off_R + bR + 255) & ~(size_t)255,
           off_si = (off_ti + bI + 255) & ~(size_t)255, off_Q = (off_si + bI + 255) & ~(size_t)255,
           off_Fo = (off_Q + bQ_ + 255) & ~(size_t)255, off_Fc = (off_Fo + bFo + 255) & ~(size_t)255,
           total = off_Fc + bFc;
    ndtgpu_status rc = ts->ensure_stage(total);
    if (rc != NDTGPU_OK) return rc;
    char *base = (char *)ts->stage;
    HIP_TRY(hipMemcpyAsync(base, T16, bT, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(base + off_ti, tidx, bI, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(base + off_si, sidx, bI, hipMemcpyHostToDevice, st));
    if (Q36) HIP_TRY(hipMemcpyAsync(base + off_Q, Q36, n_pairs * 36 * sizeof(double), hipMemcpyHostToDevice, st));
    if (feat_off) {
        HIP_TRY(hipMemcpyAsync(base + off_Fo, feat_off, bFo, hipMemcpyHostToDevice, st));
        if (bFc) HIP_TRY(hipMemcpyAsync(base + off_Fc, feat_cells, bFc, hipMemcpyHostToDevice, st));
    }
    rc = match_device_core(ts, (const uint32_t *)(base + off_ti), ss, (const uint32_t *)(base + off_si), (double *)base, n_pairs, p,
                           (ndtgpu_match_result *)(base + off_R), Q36 ? (const double *)(base + off_Q) : nullptr, st,
                           feat_off ? (const unsigned *)(base + off_Fo) : nullptr, feat_off ? (const double *)(base + off_Fc) : nullptr);
    if (rc != NDTGPU_OK) return rc;
    unsigned aborted = 0;
    HIP_TRY(hipMemcpyAsync(T16, base, bT, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(results, base + off_R, bR, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&aborted, (char *)ts->work + ndt_match_abort_offset(), sizeof aborted, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (aborted) return fail(NDTGPU_ERR_HIP, "match: the persistent matcher gave up (a wave found no work for ~1 s)");
    return NDTGPU_OK;
}

static_assert(sizeof(ndtgpu_cell_record) == sizeof(NdtCell) && offsetof(ndtgpu_cell_record, slot) == offsetof(NdtCell, slot) &&
              offsetof(ndtgpu_cell_record, cov) == offsetof(NdtCell, cov), "the exchange record is the device cell record");

size_t ndtgpu_mapset_pack_bytes(const ndtgpu_mapset *s, uint32_t cells_cap, int with_occupancy)
{
    if (!s) return 0;
    size_t b = sizeof(ndtgpu_packed_header) + (size_t)cells_cap * sizeof(NdtCell);
    if (with_occupancy) b += (size_t)s->v.grid.slots * sizeof(float);
    return (b + 15u) & ~(size_t)15u;
}

ndtgpu_status ndtgpu_mapset_pack_cells_device(ndtgpu_mapset *s, size_t first, size_t count, void *buf_dev, size_t stride,
                                              uint32_t cells_cap, int with_occupancy, ndtgpu_stream stream)
{
    if (!s || first + count > s->n_maps || (count && !buf_dev) || ((uintptr_t)buf_dev & 15u) || (stride & 15u) ||
        stride < ndtgpu_mapset_pack_bytes(s, cells_cap, with_occupancy))
        return fail(NDTGPU_ERR_INVALID, "pack_cells: bad argument (16-byte aligned buffer, stride >= ndtgpu_mapset_pack_bytes)");
    if (with_occupancy && !s->v.occ) return fail(NDTGPU_ERR_INVALID, "pack_cells: the set carries no occupancies");
    hipError_t e = ndt_launch_pack(s->v, first, count, buf_dev, stride, cells_cap, with_occupancy ? 1 : 0, 0u, (hipStream_t)stream);
    if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "pack_cells: launch", e);
    return NDTGPU_OK;
}

size_t ndtgpu_mapset_pack_bytes_sparse(const ndtgpu_mapset *s, uint32_t cells_cap, uint32_t occ_cap)
{
    if (!s) return 0;
    const size_t b = sizeof(ndtgpu_packed_header) + (size_t)cells_cap * sizeof(NdtCell) + ndt_pack_sparse_occ_bytes(occ_cap);
    return (b + 15u) & ~(size_t)15u;
}

ndtgpu_status ndtgpu_mapset_occupied_cells_max(ndtgpu_mapset *s, size_t first, size_t count, uint32_t *max_occupied,
                                               ndtgpu_stream stream)
{
    if (!s || !max_occupied || first + count > s->n_maps) return fail(NDTGPU_ERR_INVALID, "occupied_cells_max: bad argument");
    if (!s->v.occ) return fail(NDTGPU_ERR_INVALID, "occupied_cells_max: the set carries no occupancies");
    *max_occupied = 0;
    if (count == 0) return NDTGPU_OK;
    hipStream_t st = (hipStream_t)stream;
    unsigned *counts_dev = nullptr;
    HIP_TRY(hipMalloc((void **)&counts_dev, count * sizeof(unsigned)));
    std::vector<unsigned> counts(count);
    hipError_t e = ndt_launch_occ_count(s->v, first, nullptr, count, counts_dev, st);
    if (e == hipSuccess) e = hipMemcpyAsync(counts.data(), counts_dev, count * sizeof(unsigned), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(counts_dev);
    if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "occupied_cells_max", e);
    for (unsigned c : counts) *max_occupied = std::max<uint32_t>(*max_occupied, c);
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_mapset_pack_cells_sparse_device(ndtgpu_mapset *s, size_t first, size_t count, void *buf_dev, size_t stride,
                                                     uint32_t cells_cap, uint32_t occ_cap, ndtgpu_stream stream)
{
    if (!s || first + count > s->n_maps || (count && !buf_dev) || ((uintptr_t)buf_dev & 15u) || (stride & 15u) ||
        stride < ndtgpu_mapset_pack_bytes_sparse(s, cells_cap, occ_cap))
        return fail(NDTGPU_ERR_INVALID, "pack_cells_sparse: bad argument (16-byte aligned buffer, stride >= ndtgpu_mapset_pack_bytes_sparse)");
    if (!s->v.occ) return fail(NDTGPU_ERR_INVALID, "pack_cells_sparse: the set carries no occupancies");
    hipError_t e = ndt_launch_pack(s->v, first, count, buf_dev, stride, cells_cap, 2, occ_cap, (hipStream_t)stream);
    if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "pack_cells_sparse: launch", e);
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_mapset_unpack_cells_device(ndtgpu_mapset *s, size_t first, size_t count, const void *buf_dev, size_t stride,
                                                int with_occupancy, ndtgpu_stream stream)
{
    if (!s || first + count > s->n_maps || (count && !buf_dev) || ((uintptr_t)buf_dev & 15u) || (stride & 15u) ||
        stride < sizeof(ndtgpu_packed_header))
        return fail(NDTGPU_ERR_INVALID, "unpack_cells: bad argument");
    if (with_occupancy && !s->v.occ) return fail(NDTGPU_ERR_INVALID, "unpack_cells: call ndtgpu_mapset_enable_occupancy first");
    hipError_t e = ndt_launch_unpack(s->v, first, count, buf_dev, stride, with_occupancy ? 1 : 0, (hipStream_t)stream);
    if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "unpack_cells: launch", e);
    return s->touch((hipStream_t)stream);
}

ndtgpu_status ndtgpu_match_aborted(ndtgpu_mapset *ts, int *aborted)
{
    if (!ts || !aborted) return fail(NDTGPU_ERR_INVALID, "match_aborted: bad argument");
    *aborted = 0;
    if (!ts->work) return NDTGPU_OK;                     // no persistent launch has used this set as a target
    { ndtgpu_status wrc_ = ts->wait_all(); if (wrc_ != NDTGPU_OK) return wrc_; }
    unsigned w = 0;
    HIP_TRY(hipMemcpy(&w, (char *)ts->work + ndt_match_abort_offset(), sizeof w, hipMemcpyDeviceToHost));
    *aborted = w != 0u;
    return NDTGPU_OK;
}

// Small batches: the host runs the Newton / More-Thuente state machine (the same ndt_solver.h code the
// persistent kernel runs on the device) and every derivative evaluation is one multi-workgroup kernel,
// so a single registration uses the whole chip instead of one CU.  Used below NDTGPU_HOST_LOOP_MAX pairs.
static ndtgpu_status match_host_driven(ndtgpu_mapset *ts, const uint32_t *tidx, ndtgpu_mapset *ss, const uint32_t *sidx,
                                       double *T16, size_t n_pairs, const NdtMatchParamsDev &p, const double *Q36,
                                       ndtgpu_match_result *results, hipStream_t st)
{
    const unsigned max_groups = 128;
    ndtgpu_status rc = ts->ensure_stage(max_groups * 32 * sizeof(double));
    if (rc != NDTGPU_OK) return rc;
    double *partials_dev = (double *)ts->stage;
    std::vector<double> partials(max_groups * 32);
    for (size_t k = 0; k < n_pairs; k++) {
        long long terms_g = 0, terms_h = 0;
        NdtMapCounters cs, ct;
        HIP_TRY(hipMemcpy(&cs, ss->v.counters + sidx[k], sizeof cs, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&ct, ts->v.counters + tidx[k], sizeof ct, hipMemcpyDeviceToHost));
        unsigned groups = (cs.n_cells + 511u) / 512u;
        if (groups < 1) groups = 1;
        if (groups > max_groups) groups = max_groups;
        ts->ev_valid[1] = false;
        MatchState ms;
        NewtonWs ws;
        match_state_init(ms, T16 + 16 * k, p, Q36 ? Q36 + 36 * k : nullptr);
        while (!ms.done) {
            hipError_t e = ndt_launch_eval(ts->v, tidx[k], ss->v, sidx[k], ms.Teval, p.n_neighbours, ms.with_h, p.lfd1,
                                           p.lfd2, groups, partials_dev, st);
            if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "match: eval launch", e);
            HIP_TRY(hipMemcpyAsync(partials.data(), partials_dev, groups * 32 * sizeof(double), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            double sums[29];
            for (int q = 0; q < 29; q++) {
                double s = 0;
                for (unsigned g = 0; g < groups; g++) s += partials[g * 32 + q];
                sums[q] = s;
            }
            if (ms.with_h) terms_h += (long long)sums[28]; else terms_g += (long long)sums[28];
            if (getenv("NDTGPU_TRACE"))      // (debugging aid of the host-driven loop: one line per evaluation)
                fprintf(stderr, "hip eval with_h %d phase %d itr %d nfev %d score %.17g g %.9e %.9e %.9e\n", ms.with_h, ms.phase, ms.itr_ctr, ms.mt.nfev,
                        sums[0], sums[1], sums[2], sums[6]);
            match_state_step(ms, sums, p, ws);
        }
        NdtMatchResultDev o;
        match_state_result(ms, T16 + 16 * k, o);
        o.n_source = (int32_t)cs.n_cells;
        o.n_target = (int32_t)ct.n_cells;
        o.cycles_eval = 0;
        o.cycles_solver = 0;
        o.pair_terms_g = terms_g;
        o.pair_terms_h = terms_h;
        memcpy(&results[k], &o, sizeof o);
    }
    return NDTGPU_OK;
}

// Host-pointer batches that cannot fill the chip with one workgroup per registration (the reference's one-link-at-a-time
// call is the extreme case).  Returns NDTGPU_OK with *done = false when the persistent kernel is the better shape.
static ndtgpu_status match_coop(ndtgpu_mapset *ts, const uint32_t *tidx, ndtgpu_mapset *ss, const uint32_t *sidx,
                                double *T16, size_t n_pairs, const NdtMatchParamsDev &p, const double *Q36,
                                ndtgpu_match_result *results, hipStream_t st, bool *done)
{
    *done = false;
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
        n_cu = 256;
    if (n_pairs > (size_t)n_cu / 2) return NDTGPU_OK;
    const char *coop_env = getenv("NDTGPU_COOP");             // NDTGPU_COOP=0: persistent kernel above 8 pairs, =1: never (A/B)
    if (coop_env && atoi(coop_env) == 0 && n_pairs > NDTGPU_HOST_LOOP_MAX) return NDTGPU_OK;
    // More than a handful of registrations on a set of small maps (fewer than 16 k cells per map, i.e. 2D scans: four
    // chunks each): one CU per registration is as fast or faster (2D, 100 k points: 16 / 64 / 128 pairs 1.47 / 1.59 / 2.39 ms
    // here against 1.46 / 1.47 / 1.48 ms on the persistent kernel; 12 k-cell 3D maps: 4.9 / 12.8 ms against 46 ms).
    // (NDTGPU_COOP=1 takes the grid-barrier matcher regardless)
    if (n_pairs > NDTGPU_HOST_LOOP_MAX && ss->v.grid.max_cells < NDTGPU_COOP_MIN_SET_CELLS && !(coop_env && atoi(coop_env) == 1)) return NDTGPU_OK;
    CoopPlan pl;
    if (!coop_plan(ss, n_pairs, p, pl)) return NDTGPU_OK;
    // One pinned host block mirrors the device staging block [T | results | target idx | source idx | Q], followed by the
    // control words read back at the end: one copy in, the launch, one copy of poses + results and the control words out,
    // ONE wait.  (Round 2: eight pageable copies and four waits -- a third of a single-pair call.)
    const size_t bT = n_pairs * 16 * sizeof(double), bR = n_pairs * sizeof(ndtgpu_match_result), bI = n_pairs * sizeof(uint32_t);
    const size_t off_R = (bT + 255) & ~(size_t)255, off_ti = (off_R + bR + 255) & ~(size_t)255,
                 off_si = (off_ti + bI + 255) & ~(size_t)255, off_Q = (off_si + bI + 255) & ~(size_t)255;
    const size_t total = off_Q + (Q36 ? n_pairs * 36 * sizeof(double) : 0);
    const size_t off_ctrl = (total + 255) & ~(size_t)255;
    ndtgpu_status rc = ts->ensure_stage(total);
    if (rc != NDTGPU_OK) return rc;
    rc = ts->ensure_pin(off_ctrl + n_pairs * 16 + n_pairs * sizeof(unsigned));
    if (rc != NDTGPU_OK) return rc;
    char *base = (char *)ts->stage, *hp = (char *)ts->pin;
    memcpy(hp, T16, bT);
    memcpy(hp + off_ti, tidx, bI);
    memcpy(hp + off_si, sidx, bI);
    if (Q36) memcpy(hp + off_Q, Q36, n_pairs * 36 * sizeof(double));
    const unsigned *ctrl = reinterpret_cast<const unsigned *>(hp + off_ctrl);
    std::vector<double> Tin(T16, T16 + 16 * n_pairs);          // (the poses as they came in: a registration that has to be re-run)
    {
        std::lock_guard<std::mutex> coop_lock(g_coop_mutex);
        rc = ts->ensure_coop(n_pairs * pl.stride + (pl.pool ? ndt_match_pool_ctrl_bytes() : 0));
        if (rc != NDTGPU_OK) return rc;
        // The grid-barrier kernel (up to eight registrations) reads poses, indices and Tcov from the pinned block where it is and
        // writes poses and results there: 16 doubles per workgroup over the link at the start, 192 bytes back at the end, instead of
        // two copies on the stream (~6 us each) around a 0.2 ms launch.  The task pool keeps its staging copies.
        const bool direct = !pl.pool && env_int("NDTGPU_COOP_DIRECT", 1) != 0;
        // ... and the host then watches one word per registration in that block, which workgroup 0 sets behind pose and result,
        // instead of waiting for the stream (the runtime's completion signal costs ~8 us more than the store takes to arrive); the
        // next launch of this kind is ordered behind the kernel's end by its event, like any asynchronous one
        const bool poll = direct && env_int("NDTGPU_COOP_POLL", 1) != 0;
        unsigned *flags = reinterpret_cast<unsigned *>(hp + off_ctrl + n_pairs * 16);
        if (poll) for (size_t k = 0; k < n_pairs; k++) __atomic_store_n(&flags[k], 0u, __ATOMIC_RELEASE);
        if (direct) base = hp;
        else HIP_TRY(hipMemcpyAsync(base, hp, total, hipMemcpyHostToDevice, st));
        // only blocks this set has not seen finish cleanly at this stride are cleared
        const bool clear = ts->coop_clean_stride != pl.stride || ts->coop_clean_upto < n_pairs;
        const size_t clean_before = clear ? n_pairs : ts->coop_clean_upto;
        ts->coop_clean_stride = pl.stride;
        ts->coop_clean_upto = 0;                                // (until this call is known to have ended cleanly)
        rc = coop_enqueue(ts, ss, (const uint32_t *)(base + off_ti), (const uint32_t *)(base + off_si), (double *)base,
                          reinterpret_cast<NdtMatchResultDev *>(base + off_R), Q36 ? (const double *)(base + off_Q) : nullptr,
                          n_pairs, p, pl, clear, poll, st, poll ? flags : nullptr);
        if (rc != NDTGPU_OK) return rc;
        if (!direct) HIP_TRY(hipMemcpyAsync(hp, base, off_R + bR, hipMemcpyDeviceToHost, st));         // poses + results
        bool seen = poll;
        if (poll) {
            const auto t_poll = std::chrono::steady_clock::now();
            for (size_t k = 0; k < n_pairs && seen; k++) {
                unsigned spins = 0;
                while (__atomic_load_n(&flags[k], __ATOMIC_ACQUIRE) == 0u) {
                    __builtin_ia32_pause();
                    if ((++spins & 0xFFFFu) == 0u && std::chrono::steady_clock::now() - t_poll > std::chrono::seconds(2)) { seen = false; break; }
                }
            }
        }
        if (!seen) {
            HIP_TRY(hipStreamSynchronize(st));
            g_coop_ev_valid = false;      // (this stream waited for the last asynchronous launch, and is drained now)
        }
        {                                 // a registration the launch gave up on reports exit code -4 (both kernels)
            const ndtgpu_match_result *hr = reinterpret_cast<const ndtgpu_match_result *>(hp + off_R);
            unsigned *cw = reinterpret_cast<unsigned *>(hp + off_ctrl);
            for (size_t k = 0; k < n_pairs; k++) { cw[4 * k] = 0u; cw[4 * k + 1] = hr[k].exit_code == -4 ? 1u : 0u; }
        }
        bool any_bad = false;
        for (size_t k = 0; k < n_pairs; k++) any_bad = any_bad || ctrl[4 * k + 1] != 0u;
        ts->coop_clean_upto = any_bad ? 0 : clean_before;
    }
    ts->ev_valid[1] = false;          // (ndtgpu_last_kernel_ms(1): no persistent launch was timed by this call)
    memcpy(T16, hp, bT);
    memcpy(results, hp + off_R, bR);
    // A registration whose grid barrier gave up (it cannot with a co-resident grid; the bounded spin stays as a guard
    // against a foreign kernel holding CUs) is run again on the persistent kernel: the call does not fail.
    std::vector<size_t> bad;
    for (size_t k = 0; k < n_pairs; k++)
        if (ctrl[4 * k + 1]) bad.push_back(k);
    if (!bad.empty()) {
        std::vector<double> keep(16 * bad.size());
        for (size_t j = 0; j < bad.size(); j++) memcpy(&keep[16 * j], &Tin[16 * bad[j]], 16 * sizeof(double));
        Tin.swap(keep);
    }
    if (!bad.empty()) {
        std::vector<uint32_t> bt(bad.size()), bs(bad.size());
        std::vector<double> bQ;
        std::vector<ndtgpu_match_result> br(bad.size());
        for (size_t j = 0; j < bad.size(); j++) { bt[j] = tidx[bad[j]]; bs[j] = sidx[bad[j]]; }
        if (Q36) {
            bQ.resize(36 * bad.size());
            for (size_t j = 0; j < bad.size(); j++) memcpy(&bQ[36 * j], Q36 + 36 * bad[j], 36 * sizeof(double));
        }
        rc = match_persistent_host(ts, bt.data(), ss, bs.data(), Tin.data(), bad.size(), p, Q36 ? bQ.data() : nullptr, br.data(), st);
        if (rc != NDTGPU_OK) return rc;
        for (size_t j = 0; j < bad.size(); j++) {
            memcpy(T16 + 16 * bad[j], &Tin[16 * j], 16 * sizeof(double));
            results[bad[j]] = br[j];
        }
    }
    *done = true;
    return NDTGPU_OK;
}

static ndtgpu_status match_batch_common(ndtgpu_mapset *ts, const uint32_t *tidx, ndtgpu_mapset *ss, const uint32_t *sidx,
                                        double *T16, size_t n_pairs, const ndtgpu_match_params *prm, const double *Q36,
                                        int fusion_flags, ndtgpu_match_result *results, ndtgpu_stream stream);

ndtgpu_status ndtgpu_match_batch(ndtgpu_mapset *ts, const uint32_t *tidx, ndtgpu_mapset *ss, const uint32_t *sidx,
                                 double *T16, size_t n_pairs, const ndtgpu_match_params *prm,
                                 ndtgpu_match_result *results, ndtgpu_stream stream)
{
    return match_batch_common(ts, tidx, ss, sidx, T16, n_pairs, prm, nullptr, 0, results, stream);
}

// 6x6 inverse by Gauss-Jordan with partial pivoting (Eigen: Tcov.inverse(), fusion.h:845)
static bool invert6(const double *A, double *inv)
{
    double a[6][12];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) { a[i][j] = A[i * 6 + j]; a[i][6 + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 6; c++) {
        int piv = c;
        for (int r = c + 1; r < 6; r++)
            if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (a[piv][c] == 0.0) return false;
        if (piv != c)
            for (int j = 0; j < 12; j++) std::swap(a[c][j], a[piv][j]);
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
    return true;
}

ndtgpu_status ndtgpu_match_fusion_batch(ndtgpu_mapset *ts, const uint32_t *tidx, ndtgpu_mapset *ss, const uint32_t *sidx,
                                        double *T16, const double *Tcov36, size_t n_pairs,
                                        const ndtgpu_match_params *prm, int use_soft_constraints,
                                        ndtgpu_match_result *results, ndtgpu_stream stream)
{
    if (use_soft_constraints < 0 || use_soft_constraints > 3)
        return fail(NDTGPU_ERR_INVALID, "match_fusion: use_soft_constraints is a 2-bit set (bit 0 useSoftConstraints, bit 1 useTikhonovRegularization)");
    const int flags = use_soft_constraints;       // bit 0 useSoftConstraints, bit 1 useTikhonovRegularization
    if (!flags) return match_batch_common(ts, tidx, ss, sidx, T16, n_pairs, prm, nullptr, 0, results, stream);
    if (!Tcov36) return fail(NDTGPU_ERR_INVALID, "match_fusion: Tcov missing");
    std::vector<double> Q(36 * n_pairs);
    for (size_t k = 0; k < n_pairs; k++)
        if (!invert6(Tcov36 + 36 * k, Q.data() + 36 * k)) return fail(NDTGPU_ERR_INVALID, "match_fusion: singular Tcov");
    return match_batch_common(ts, tidx, ss, sidx, T16, n_pairs, prm, Q.data(), flags, results, stream);
}

ndtgpu_status ndtgpu_match_fusion_feat_batch(ndtgpu_mapset *ts, const uint32_t *tidx, ndtgpu_mapset *ss, const uint32_t *sidx,
                                             double *T16, const double *Tcov36, const ndtgpu_feat_pairs *feat, size_t n_pairs,
                                             const ndtgpu_match_params *prm, int flags, ndtgpu_match_result *results,
                                             ndtgpu_stream stream)
{
    if (flags < 0 || flags > 7) return fail(NDTGPU_ERR_INVALID, "match_fusion_feat: flags is a 3-bit set");
    if (!feat || !feat->offsets)   // no feature maps: bit 2 (the joint line search of the feature maps) has nothing to act on
        return ndtgpu_match_fusion_batch(ts, tidx, ss, sidx, T16, Tcov36, n_pairs, prm, flags & 3, results, stream);
    if (!ts || !ss || (n_pairs && (!tidx || !sidx || !T16 || !results)))
        return fail(NDTGPU_ERR_INVALID, "match_fusion_feat: bad argument");
    if (n_pairs == 0) return NDTGPU_OK;
    for (size_t k = 0; k < n_pairs; k++) {
        if (tidx[k] >= ts->n_maps || sidx[k] >= ss->n_maps) return fail(NDTGPU_ERR_INVALID, "match_fusion_feat: map index");
        if (feat->offsets[k + 1] < feat->offsets[k]) return fail(NDTGPU_ERR_INVALID, "match_fusion_feat: offsets must not decrease");
        if (feat->offsets[k + 1] - feat->offsets[k] > 64u)
            return fail(NDTGPU_ERR_CAPACITY, "match_fusion_feat: at most 64 correspondences per registration");
    }
    const size_t total = feat->offsets[n_pairs];
    if (total && (!feat->src_mean || !feat->src_cov || !feat->tgt_mean || !feat->tgt_cov))
        return fail(NDTGPU_ERR_INVALID, "match_fusion_feat: cell arrays missing");
    std::vector<double> Q;
    if (flags & 3) {
        if (!Tcov36) return fail(NDTGPU_ERR_INVALID, "match_fusion_feat: Tcov missing");
        Q.resize(36 * n_pairs);
        for (size_t k = 0; k < n_pairs; k++)
            if (!invert6(Tcov36 + 36 * k, Q.data() + 36 * k)) return fail(NDTGPU_ERR_INVALID, "match_fusion_feat: singular Tcov");
    }
    std::vector<double> cells(total * 18);
    for (size_t i = 0; i < total; i++) {
        double *c = cells.data() + 18 * i;
        for (int a = 0; a < 3; a++) { c[a] = feat->src_mean[3 * i + a]; c[9 + a] = feat->tgt_mean[3 * i + a]; }
        for (int a = 0; a < 6; a++) { c[3 + a] = feat->src_cov[6 * i + a]; c[12 + a] = feat->tgt_cov[6 * i + a]; }
    }
    hipStream_t st = (hipStream_t)stream;
    { ndtgpu_status wrc_ = ts->wait_all(); if (wrc_ != NDTGPU_OK) return wrc_; }
    { ndtgpu_status wrc_ = ss->wait_all(); if (wrc_ != NDTGPU_OK) return wrc_; }
    NdtMatchParamsDev p = to_dev(prm);
    p.fusion_flags = flags;          // bit 2 (step_control_fusion) selects lineSearchMTFusion when bit 0 is clear (fusion.h:1004)
    if (p.n_neighbours < 0 || p.n_neighbours > 3 || (p.dof_mask & 0x3f) == 0)
        return fail(NDTGPU_ERR_INVALID, "match: n_neighbours must be 0..3 and dof_mask non-empty");
    // (always the persistent matcher: the feature sums are evaluated inside its solver step)
    return match_persistent_host(ts, tidx, ss, sidx, T16, n_pairs, p, Q.empty() ? nullptr : Q.data(), results, st, feat->offsets,
                                 cells.data());
}

static ndtgpu_status match_batch_common(ndtgpu_mapset *ts, const uint32_t *tidx, ndtgpu_mapset *ss, const uint32_t *sidx,
                                        double *T16, size_t n_pairs, const ndtgpu_match_params *prm, const double *Q36,
                                        int fusion_flags, ndtgpu_match_result *results, ndtgpu_stream stream)
{
    if (!ts || !ss || (n_pairs && (!tidx || !sidx || !T16 || !results)))
        return fail(NDTGPU_ERR_INVALID, "match_batch: bad argument");
    if (n_pairs == 0) return NDTGPU_OK;
    for (size_t k = 0; k < n_pairs; k++)
        if (tidx[k] >= ts->n_maps || sidx[k] >= ss->n_maps) return fail(NDTGPU_ERR_INVALID, "match_batch: map index");
    hipStream_t st = (hipStream_t)stream;
    // builds on other streams must have finished before the maps are read (a build on THIS stream is ordered by the stream: the
    // host stages the call while it runs -- a third of the build of a single pair, tools/latency_probe.py)
    { ndtgpu_status wrc_ = ts->wait_all_on(st); if (wrc_ != NDTGPU_OK) return wrc_; }
    { ndtgpu_status wrc_ = ss->wait_all_on(st); if (wrc_ != NDTGPU_OK) return wrc_; }
    {
        NdtMatchParamsDev p = to_dev(prm);
        p.fusion_flags = fusion_flags;
        if (p.n_neighbours < 0 || p.n_neighbours > 3 || (p.dof_mask & 0x3f) == 0)
            return fail(NDTGPU_ERR_INVALID, "match: n_neighbours must be 0..3 and dof_mask non-empty");
        // NDTGPU_HOST_LOOP=1: the host runs the state machine, one launch per evaluation (A/B, debugging)
        const char *hl = getenv("NDTGPU_HOST_LOOP");
        if (n_pairs <= NDTGPU_HOST_LOOP_MAX && hl && atoi(hl))
            return match_host_driven(ts, tidx, ss, sidx, T16, n_pairs, p, Q36, results, st);
        bool done = false;
        ndtgpu_status crc = match_coop(ts, tidx, ss, sidx, T16, n_pairs, p, Q36, results, st, &done);
        if (crc != NDTGPU_OK || done) return crc;
    }
    { ndtgpu_status wrc_ = ts->wait_all(); if (wrc_ != NDTGPU_OK) return wrc_; }
    { ndtgpu_status wrc_ = ss->wait_all(); if (wrc_ != NDTGPU_OK) return wrc_; }
    NdtMatchParamsDev pp = to_dev(prm);
    pp.fusion_flags = fusion_flags;
    return match_persistent_host(ts, tidx, ss, sidx, T16, n_pairs, pp, Q36, results, st);
}

ndtgpu_status ndtgpu_covariance_batch(ndtgpu_mapset *ts, const uint32_t *tidx, ndtgpu_mapset *ss, const uint32_t *sidx,
                                      const double *T16, size_t n_links, const ndtgpu_match_params *prm, int mode,
                                      double *cov36, int32_t *singular, ndtgpu_stream stream)
{
    if (!ts || !ss || (n_links && (!tidx || !sidx || !T16 || !cov36)) || mode < 0 || mode > 1)
        return fail(NDTGPU_ERR_INVALID, "covariance: bad argument");
    if (n_links == 0) return NDTGPU_OK;
    for (size_t k = 0; k < n_links; k++)
        if (tidx[k] >= ts->n_maps || sidx[k] >= ss->n_maps) return fail(NDTGPU_ERR_INVALID, "covariance: map index");
    NdtMatchParamsDev p = to_dev(prm);
    if (p.n_neighbours < 0 || p.n_neighbours > 3) return fail(NDTGPU_ERR_INVALID, "covariance: n_neighbours must be 0..3");
    hipStream_t st = (hipStream_t)stream;
    { ndtgpu_status wrc_ = ts->wait_all(); if (wrc_ != NDTGPU_OK) return wrc_; }
    { ndtgpu_status wrc_ = ss->wait_all(); if (wrc_ != NDTGPU_OK) return wrc_; }
    const size_t bT = n_links * 16 * sizeof(double), bI = n_links * sizeof(uint32_t), bC = n_links * 36 * sizeof(double);
    const size_t off_t = (bT + 255) & ~(size_t)255, off_s = (off_t + bI + 255) & ~(size_t)255,
                 off_c = (off_s + bI + 255) & ~(size_t)255, off_f = (off_c + bC + 255) & ~(size_t)255;
    ndtgpu_status rc = ts->ensure_stage(off_f + n_links * sizeof(int));
    if (rc != NDTGPU_OK) return rc;
    char *base = (char *)ts->stage;
    HIP_TRY(hipMemcpyAsync(base, T16, bT, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(base + off_t, tidx, bI, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(base + off_s, sidx, bI, hipMemcpyHostToDevice, st));
    hipError_t e = ndt_launch_covariance(ts->v, (const uint32_t *)(base + off_t), ss->v, (const uint32_t *)(base + off_s),
                                         (const double *)base, n_links, p.n_neighbours, p.lfd1, p.lfd2, mode,
                                         (double *)(base + off_c), (int *)(base + off_f), st);
    if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "covariance: launch", e);
    HIP_TRY(hipMemcpyAsync(cov36, base + off_c, bC, hipMemcpyDeviceToHost, st));
    if (singular) HIP_TRY(hipMemcpyAsync(singular, base + off_f, n_links * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_match_d2d(ndtgpu_mapset *ts, size_t tmap, ndtgpu_mapset *ss, size_t smap, double T16[16],
                               const ndtgpu_match_params *prm, ndtgpu_match_result *result)
{
    uint32_t ti = (uint32_t)tmap, si = (uint32_t)smap;
    return ndtgpu_match_batch(ts, &ti, ss, &si, T16, 1, prm, result, nullptr);
}


// ---- the fuser bank: NDTFeatureFuserHMT::update for a batch of independent fusers as ONE call (include/ndtgpu.h) ----------
// Per call and slot: [host] the odometry model, the soft-constraint covariance, the odometry cells, the scan frame and the scan
// map's centre (ndtgpu_fuser_prepare) -> ONE upload -> [device] scan into the node map's frame, scan map build, matchFusion
// against the slot's node map, the matcher's covariance, the post-registration step (pose, gates, accumulated covariance),
// scan into the frame of the new pose, ray-traced fuse-in.  No host round trip in between; the host looks at the poses when it
// asks for them (ndtgpu_fuser_poses) or at the next call, which needs them.

struct ndtgpu_fuser_bank {
    ndtgpu_fuser_params prm{};
    size_t n = 0;
    ndtgpu_mapset *nodes = nullptr, *scans = nullptr;
    bool own_nodes = false;
    struct HostState {
        NdtFuserState s{};
        double Todom[16];
        bool is_init = false;
    };
    std::vector<HostState> st;
    NdtFuserState *st_dev = nullptr;
    double *sensor_pose_dev = nullptr;
    // staging of one call: a pinned block and its device twin
    //   [count] Tscan16 | Tmotion16 | Test16 | Q36 | origins3 | centres3 | feat cells (40 x 18) | feat offsets | idx | spose16 | fuse origins3 |
    //   match results | cov36 | cov flags | results
    char *pin = nullptr, *dev = nullptr;
    size_t stage_bytes = 0;
    float *xyz_a = nullptr, *xyz_b = nullptr;     // the scans in the node frame before / after the registration (packed xyz)
    size_t xyz_cap = 0;
    void *xyz_in = nullptr;                       // host clouds (the *_host entries) on their way in
    size_t xyz_in_bytes = 0;
    hipStream_t own_st = nullptr;
    hipEvent_t done_ev = nullptr;
    bool in_flight = false;
    size_t fl_first = 0, fl_count = 0;
    size_t off_res = 0, off_pin_res = 0;          // where the in-flight call's results sit in the staging block
    hipStream_t fl_stream = nullptr;
};

namespace {
struct FuserLayout {
    size_t Tscan, Tmotion, Test, Q, origin, centre, feat, foff, idx, spose, forigin, match, cov, covflag, res, total;
};
FuserLayout fuser_layout(size_t count, bool feat)
{
    FuserLayout L;
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t o = at; at = (at + bytes + 255) & ~(size_t)255; return o; };
    L.Tscan = take(count * 16 * sizeof(double));
    L.Tmotion = take(count * 16 * sizeof(double));
    L.Test = take(count * 16 * sizeof(double));
    L.Q = take(count * 36 * sizeof(double));
    L.origin = take(count * 3 * sizeof(double));
    L.centre = take(count * 3 * sizeof(double));
    L.feat = take(feat ? count * 40 * 18 * sizeof(double) : 0);
    L.foff = take((count + 1) * sizeof(uint32_t));
    L.idx = take(count * sizeof(uint32_t));
    L.spose = take(count * 16 * sizeof(double));
    L.forigin = take(count * 3 * sizeof(double));
    L.match = take(count * sizeof(NdtMatchResultDev));
    L.cov = take(count * 36 * sizeof(double));
    L.covflag = take(count * sizeof(int));
    L.res = take(count * sizeof(NdtFuserResultDev));
    L.total = at;
    return L;
}
void pose_identity(double *T) { for (int q = 0; q < 16; q++) T[q] = (q % 5 == 0) ? 1.0 : 0.0; }
}  // namespace

static_assert(sizeof(ndtgpu_fuser_result) == sizeof(NdtFuserResultDev), "fuser result layouts must agree");

void ndtgpu_default_fuser_params(ndtgpu_fuser_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof *p);
    // NDTFeatureFuserHMT::Params() (ndt_feature_fuser_hmt.h:58-101)
    p->resolution = 1.0;
    p->map_size_x = 40.0; p->map_size_y = 40.0; p->map_size_z = 10.0;
    p->sensor_range = 3.0;
    p->max_translation_norm = 1.0;
    p->max_rotation_norm = M_PI / 4.0;
    p->check_consistency = 0;
    p->fuse_incomplete = 0;
    p->use_odom = 1;
    p->neighbours = 0;
    p->stepcontrol = 1;
    p->itr_max = 30;
    p->delta_score = 10e-4;
    p->force_odom_as_est = 0;
    p->fusion2d = 0;
    p->all_matches_valid = 0;
    p->use_soft_constraints = 1;
    p->compute_cov = 1;
    p->step_control_fusion = 1;
    p->use_tikhonov = 1;
    p->covariance_mode = 0;
    // MotionModel2d::Params() (motion_model.hpp:128-136)
    p->motion_Cd = 0.001; p->motion_Ct = 0.001; p->motion_Dd = 0.005; p->motion_Dt = 0.005; p->motion_Td = 0.001; p->motion_Tt = 0.001;
    pose_identity(p->sensor_pose);
    p->max_cells = 0;
}

ndtgpu_status ndtgpu_fuser_prepare(const ndtgpu_fuser_params *prm, const double Tnow16[16], const double Tmotion16[16],
                                   const double node_centre[3], ndtgpu_fuser_prepared *out)
{
    if (!prm || !Tnow16 || !Tmotion16 || !node_centre || !out) return fail(NDTGPU_ERR_INVALID, "fuser_prepare: null argument");
    if (!(prm->resolution > 0)) return fail(NDTGPU_ERR_INVALID, "fuser_prepare: resolution");
    // fuser_hmt.cpp:124-146 -- the odometry "constraints" (MotionModel2d::getMeasurementCov, motion_model.cpp:190-207)
    double e3[3];
    ndt_euler012(Tmotion16, e3);
    const double rx = Tmotion16[12], ry = Tmotion16[13], rot = e3[2];
    const double dist = std::sqrt(rx * rx + ry * ry);
    const double R00 = prm->motion_Dd * dist * dist + prm->motion_Dt * rot * rot;
    const double R11 = prm->motion_Cd * dist * dist + prm->motion_Ct * rot * rot;
    const double R22 = prm->motion_Td * dist * dist + prm->motion_Tt * rot * rot;
    for (double &v : out->odom_cov) v = 0.0;
    out->odom_cov[0] = R00; out->odom_cov[4] = R11;
    out->odom_cov[8] = 0.01;                       // "the height in the ndt feature vec and not rotational variance"
    for (double &v : out->Tcov) v = 0.0;
    for (int a = 0; a < 6; a++) out->Tcov[a * 6 + a] = 1.0;
    out->Tcov[0] = R00; out->Tcov[7] = R11; out->Tcov[35] = R22;     // getCovMatrix6; (2,2) = (3,3) = (4,4) = 1 (:144-146)
    // :166-190 (globalTransf): the scan goes to the node map's frame by Tinit * sensor_pose, Tinit = Tnow
    ndt_pose_mul(Tnow16, prm->sensor_pose, out->Tscan);
    for (int a = 0; a < 3; a++) {
        out->range_origin[a] = out->Tscan[12 + a];
        // loadPointCloudCentroid (:201-202): the scan map's centre on the lattice of the node map's
        const double diff = out->range_origin[a] - node_centre[a];
        out->scan_centre[a] = node_centre[a] + std::floor(diff / prm->resolution) * prm->resolution;
    }
    // :291-334 -- the odometry cells: a pair (previous pose + motion | current pose), both moved into the node map's frame
    // by Tnow (pseudoTransformNDTMap: mean' = T mean, cov' = R cov R^T); the LAST current cell keeps the un-rotated covariance
    double RC[9], RCRt[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s2 = 0;
            for (int k = 0; k < 3; k++) s2 += Tnow16[k * 4 + i] * out->odom_cov[k * 3 + j];
            RC[i * 3 + j] = s2;
        }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s2 = 0;
            for (int k = 0; k < 3; k++) s2 += RC[i * 3 + k] * Tnow16[k * 4 + j];
            RCRt[i * 3 + j] = s2;
        }
    const int ij[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
    for (int k = 0; k < 6; k++) {
        out->feat_cov_rotated[k] = RCRt[ij[k][0] * 3 + ij[k][1]];
        out->feat_cov_plain[k] = out->odom_cov[ij[k][0] * 3 + ij[k][1]];
    }
    for (int a = 0; a < 3; a++) {
        out->feat_src_mean[a] = Tnow16[12 + a];                                                        // Tinit * 0
        out->feat_tgt_mean[a] = Tnow16[a] * Tmotion16[12] + Tnow16[4 + a] * Tmotion16[13] + Tnow16[8 + a] * Tmotion16[14] + Tnow16[12 + a];
    }
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_fuser_bank_destroy(ndtgpu_fuser_bank *b)
{
    if (!b) return NDTGPU_OK;
    if (b->in_flight && b->done_ev) (void)hipEventSynchronize(b->done_ev);
    if (b->done_ev) (void)hipEventDestroy(b->done_ev);
    if (b->scans) (void)ndtgpu_mapset_destroy(b->scans);
    if (b->own_nodes && b->nodes) (void)ndtgpu_mapset_destroy(b->nodes);
    if (b->st_dev) (void)hipFree(b->st_dev);
    if (b->sensor_pose_dev) (void)hipFree(b->sensor_pose_dev);
    if (b->pin) (void)hipHostFree(b->pin);
    if (b->dev) (void)hipFree(b->dev);
    if (b->xyz_a) (void)hipFree(b->xyz_a);
    if (b->xyz_b) (void)hipFree(b->xyz_b);
    if (b->xyz_in) (void)hipFree(b->xyz_in);
    if (b->own_st) { (void)hipStreamSynchronize(b->own_st); (void)hipStreamDestroy(b->own_st); }
    delete b;
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_fuser_bank_create(const ndtgpu_fuser_params *prm, size_t n_fusers, ndtgpu_mapset *node_maps, ndtgpu_fuser_bank **out)
{
    if (!prm || !out || n_fusers == 0) return fail(NDTGPU_ERR_INVALID, "fuser_bank_create: bad argument");
    if (!(prm->resolution > 0) || !(prm->sensor_range > 0) || prm->neighbours < 0 || prm->neighbours > 3 || prm->itr_max < 0)
        return fail(NDTGPU_ERR_INVALID, "fuser_bank_create: resolution / sensor_range must be positive, neighbours 0..3");
    if (!have_device()) return fail(NDTGPU_ERR_NO_DEVICE, "fuser_bank_create: no HIP device");
    if (node_maps && node_maps->n_maps < n_fusers) return fail(NDTGPU_ERR_INVALID, "fuser_bank_create: node_maps holds fewer maps than fusers");
    if (node_maps && node_maps->v.grid.res != prm->resolution) return fail(NDTGPU_ERR_INVALID, "fuser_bank_create: node_maps has another cell size");
    ndtgpu_fuser_bank *b = new (std::nothrow) ndtgpu_fuser_bank();
    if (!b) return fail(NDTGPU_ERR_ALLOC, "fuser_bank_create: host alloc");
    b->prm = *prm;
    b->n = n_fusers;
    b->st.resize(n_fusers);
    for (auto &h : b->st) { pose_identity(h.s.Tnow); pose_identity(h.s.Tlast_fuse); pose_identity(h.Todom); }
    ndtgpu_status rc = NDTGPU_OK;
    if (node_maps) {
        b->nodes = node_maps;
    } else {
        ndtgpu_grid_params g{};
        g.res = prm->resolution;
        g.size[0] = prm->map_size_x; g.size[1] = prm->map_size_y; g.size[2] = prm->map_size_z;
        g.max_cells = prm->max_cells;
        rc = ndtgpu_mapset_create(&g, n_fusers, &b->nodes);
        b->own_nodes = rc == NDTGPU_OK;
    }
    if (rc == NDTGPU_OK) rc = ndtgpu_mapset_enable_occupancy(b->nodes);
    if (rc == NDTGPU_OK) {
        // the scan maps: localMapSize (ndt_feature_fuser_hmt.h:224-226), centres set per update
        ndtgpu_grid_params g{};
        g.res = prm->resolution;
        g.size[0] = g.size[1] = prm->sensor_range + 3.0 * prm->resolution;
        g.size[2] = prm->map_size_z;
        g.max_cells = prm->max_cells;
        rc = ndtgpu_mapset_create(&g, n_fusers, &b->scans);
    }
    hipError_t e = hipSuccess;
    if (rc == NDTGPU_OK) e = hipMalloc((void **)&b->st_dev, n_fusers * sizeof(NdtFuserState));
    if (rc == NDTGPU_OK && e == hipSuccess) e = hipMalloc((void **)&b->sensor_pose_dev, 16 * sizeof(double));
    if (rc == NDTGPU_OK && e == hipSuccess) e = hipMemcpy(b->sensor_pose_dev, prm->sensor_pose, 16 * sizeof(double), hipMemcpyHostToDevice);
    if (rc == NDTGPU_OK && e == hipSuccess) e = hipEventCreateWithFlags(&b->done_ev, hipEventDisableTiming);
    if (rc != NDTGPU_OK || e != hipSuccess) {
        const std::string why = rc != NDTGPU_OK ? g_err : std::string("fuser_bank_create: ") + hipGetErrorString(e);
        ndtgpu_fuser_bank_destroy(b);
        return fail(rc != NDTGPU_OK ? rc : NDTGPU_ERR_HIP, why.c_str());
    }
    *out = b;
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_fuser_bank_mapsets(ndtgpu_fuser_bank *b, ndtgpu_mapset **node_maps, ndtgpu_mapset **scan_maps)
{
    if (!b) return fail(NDTGPU_ERR_INVALID, "fuser_bank_mapsets: null");
    if (node_maps) *node_maps = b->nodes;
    if (scan_maps) *scan_maps = b->scans;
    return NDTGPU_OK;
}

// the host's copy of the pose state catches up with the device: waits for the call in flight
static ndtgpu_status fuser_catch_up(ndtgpu_fuser_bank *b)
{
    if (!b->in_flight) return NDTGPU_OK;
    HIP_TRY(hipEventSynchronize(b->done_ev));
    b->in_flight = false;
    if (b->fl_count) {
        std::vector<NdtFuserState> tmp(b->fl_count);
        HIP_TRY(hipMemcpy(tmp.data(), b->st_dev + b->fl_first, b->fl_count * sizeof(NdtFuserState), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < b->fl_count; k++) b->st[b->fl_first + k].s = tmp[k];
    }
    return NDTGPU_OK;
}

static ndtgpu_status fuser_stage(ndtgpu_fuser_bank *b, size_t bytes, size_t count, size_t n_points)
{
    if (bytes > b->stage_bytes) {
        if (b->pin) (void)hipHostFree(b->pin);
        if (b->dev) (void)hipFree(b->dev);
        b->pin = b->dev = nullptr;
        b->stage_bytes = 0;
        HIP_TRY(hipHostMalloc((void **)&b->pin, bytes, hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **)&b->dev, bytes));
        b->stage_bytes = bytes;
    }
    const size_t need = count * n_points * 3;
    if (need > b->xyz_cap) {
        if (b->xyz_a) (void)hipFree(b->xyz_a);
        if (b->xyz_b) (void)hipFree(b->xyz_b);
        b->xyz_a = b->xyz_b = nullptr;
        b->xyz_cap = 0;
        HIP_TRY(hipMalloc((void **)&b->xyz_a, need * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&b->xyz_b, need * sizeof(float)));
        b->xyz_cap = need;
    }
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_fuser_initialize_batch(ndtgpu_fuser_bank *b, size_t first, size_t count, const double *initPose16,
                                            const void *xyz_dev, size_t n_points, size_t stride_bytes, size_t map_stride_bytes,
                                            ndtgpu_stream stream)
{
    if (!b || first + count > b->n || (count && (!initPose16 || (n_points && !xyz_dev))) || stride_bytes < 12 || (stride_bytes & 3) ||
        n_points > 0xFFFFFFFFull)
        return fail(NDTGPU_ERR_INVALID, "fuser_initialize: bad argument");
    if (count == 0) return NDTGPU_OK;
    ndtgpu_status rc = fuser_catch_up(b);
    if (rc != NDTGPU_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const FuserLayout L = fuser_layout(count, false);
    rc = fuser_stage(b, L.total, count, n_points);
    if (rc != NDTGPU_OK) return rc;
    // fuser_hmt.cpp:65-102: the first cloud goes through the sensor pose, then through the initial pose (two roundings to float);
    // Tnow = initPos; the map is centred on it (z = 0) and receives the cloud from where the sensor stood
    rc = ndtgpu_mapset_clear(b->nodes, first, count);
    if (rc != NDTGPU_OK) return rc;
    double *Tinit = (double *)(b->pin + L.Tscan), *orig = (double *)(b->pin + L.forigin), *Tsens = (double *)(b->pin + L.Tmotion);
    for (size_t k = 0; k < count; k++) {
        ndtgpu_fuser_bank::HostState &h = b->st[first + k];
        const double *T0 = initPose16 + 16 * k;
        memcpy(h.s.Tnow, T0, sizeof h.s.Tnow);
        memcpy(h.s.Tlast_fuse, T0, sizeof h.s.Tlast_fuse);
        memcpy(h.Todom, T0, sizeof h.Todom);
        for (double &v : h.s.cov_mean) v = 0.0;
        for (double &v : h.s.cov) v = 0.0;
        h.is_init = true;
        const double centre[3] = {T0[12], T0[13], 0.0};
        rc = ndtgpu_mapset_set_centre(b->nodes, first + k, centre);
        if (rc != NDTGPU_OK) return rc;
        memcpy(Tinit + 16 * k, T0, 16 * sizeof(double));
        memcpy(Tsens + 16 * k, b->prm.sensor_pose, 16 * sizeof(double));
        double Ts[16];
        ndt_pose_mul(T0, b->prm.sensor_pose, Ts);              // Tnow_sensor: the origin the readings were taken from
        for (int a = 0; a < 3; a++) orig[3 * k + a] = Ts[12 + a];
    }
    HIP_TRY(hipMemcpyAsync(b->dev, b->pin, L.total, hipMemcpyHostToDevice, st));
    {
        std::vector<NdtFuserState> tmp(count);
        for (size_t k = 0; k < count; k++) tmp[k] = b->st[first + k].s;
        HIP_TRY(hipMemcpy(b->st_dev + first, tmp.data(), count * sizeof(NdtFuserState), hipMemcpyHostToDevice));
    }
    hipError_t e = ndt_launch_cloud_transform(xyz_dev, count, n_points, stride_bytes, map_stride_bytes, (const double *)(b->dev + L.Tmotion),
                                              (const double *)(b->dev + L.Tscan), 16, b->xyz_b, st);
    if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "fuser_initialize: transform", e);
    NdtFuseParams fp;
    fp.maxz = 100.0; fp.sensor_noise = 0.1; fp.maxnumpoints = 1e5; fp.occupancy_limit = 255.0; fp.eval_factor = 1000.0; fp.n_min = 3;   // :92-94
    e = ndt_launch_fuse(b->nodes->v, first, count, b->xyz_b, n_points, 12, n_points * 12, (const double *)(b->dev + L.forigin), fp,
                        b->nodes->nice_range(first, count), st);
    if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "fuser_initialize: fuse launch", e);
    { ndtgpu_status trc = b->nodes->touch(st); if (trc != NDTGPU_OK) return trc; }
    HIP_TRY(hipEventRecord(b->done_ev, st));
    b->in_flight = true;
    b->fl_first = first; b->fl_count = 0;                      // (the host state is already what the device holds)
    b->fl_stream = st;
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_fuser_update_batch(ndtgpu_fuser_bank *b, size_t first, size_t count, const double *Tmotion16, const void *xyz_dev,
                                        size_t n_points, size_t stride_bytes, size_t map_stride_bytes, int update_ndt_map,
                                        ndtgpu_stream stream)
{
    if (!b || first + count > b->n || (count && (!Tmotion16 || (n_points && !xyz_dev))) || stride_bytes < 12 || (stride_bytes & 3) ||
        n_points > 0xFFFFFFFFull)
        return fail(NDTGPU_ERR_INVALID, "fuser_update: bad argument");
    if (count == 0) return NDTGPU_OK;
    ndtgpu_status rc = fuser_catch_up(b);          // this call starts from the poses the previous one left
    if (rc != NDTGPU_OK) return rc;
    for (size_t k = 0; k < count; k++)
        if (!b->st[first + k].is_init) return fail(NDTGPU_ERR_INVALID, "fuser_update: call ndtgpu_fuser_initialize_batch first (NDT-FuserHMT: Call Initialize first!!)");
    const ndtgpu_fuser_params &P = b->prm;
    hipStream_t st = (hipStream_t)stream;
    const bool feat = P.use_odom != 0 && !P.fusion2d;
    const int flags = P.fusion2d ? 0 : ((P.use_soft_constraints ? 1 : 0) | (P.use_tikhonov ? 2 : 0));
    const FuserLayout L = fuser_layout(count, feat);
    rc = fuser_stage(b, L.total, count, n_points);
    if (rc != NDTGPU_OK) return rc;
    // ---- host: what depends on the odometry increment and the current pose only -----------------------------------------
    double *Tscan = (double *)(b->pin + L.Tscan), *Tm = (double *)(b->pin + L.Tmotion), *Te = (double *)(b->pin + L.Test),
           *Q = (double *)(b->pin + L.Q), *orig = (double *)(b->pin + L.origin), *cen = (double *)(b->pin + L.centre),
           *fc = (double *)(b->pin + L.feat);
    uint32_t *foff = (uint32_t *)(b->pin + L.foff), *idx = (uint32_t *)(b->pin + L.idx);
    for (size_t k = 0; k < count; k++) {
        ndtgpu_fuser_bank::HostState &h = b->st[first + k];
        const double *T = Tmotion16 + 16 * k;
        ndtgpu_fuser_prepared pp;
        rc = ndtgpu_fuser_prepare(&P, h.s.Tnow, T, &b->nodes->centres_host[(first + k) * 3], &pp);
        if (rc != NDTGPU_OK) return rc;
        memcpy(Tscan + 16 * k, pp.Tscan, sizeof pp.Tscan);
        memcpy(Tm + 16 * k, T, 16 * sizeof(double));
        memcpy(Te + 16 * k, T, 16 * sizeof(double));          // Tmotion_est starts as the odometry increment (:166-170)
        if (flags && !invert6(pp.Tcov, Q + 36 * k)) return fail(NDTGPU_ERR_INVALID, "fuser_update: singular odometry covariance");
        for (int a = 0; a < 3; a++) { orig[3 * k + a] = pp.range_origin[a]; cen[3 * k + a] = pp.scan_centre[a]; }
        if (feat) {
            for (int i = 0; i < 40; i++) {
                double *c = fc + (k * 40 + i) * 18;          // {source mean, cov | target mean, cov}: NDTMatcherFeatureD2D pairs (i, i)
                for (int a = 0; a < 3; a++) { c[a] = pp.feat_src_mean[a]; c[9 + a] = pp.feat_tgt_mean[a]; }
                for (int a = 0; a < 6; a++) { c[3 + a] = i == 39 ? pp.feat_cov_plain[a] : pp.feat_cov_rotated[a]; c[12 + a] = pp.feat_cov_rotated[a]; }
            }
        }
        foff[k] = (uint32_t)(40 * k);
        idx[k] = (uint32_t)(first + k);
        double To[16];
        ndt_pose_mul(h.Todom, T, To);                          // "we track this only for display purposes!"
        memcpy(h.Todom, To, sizeof To);
        // the scan map's centre: the launcher picks its kernel by what the host knows of the centres
        for (int a = 0; a < 3; a++) b->scans->centres_host[(first + k) * 3 + a] = pp.scan_centre[a];
        b->scans->nice_host[first + k] = ndt_grid_is_nice(b->scans->v.grid, pp.scan_centre) ? 1 : 0;
    }
    foff[count] = (uint32_t)(40 * count);
    HIP_TRY(hipMemcpyAsync(b->dev, b->pin, L.res, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b->scans->v.centres + first * 3, b->dev + L.centre, count * 3 * sizeof(double), hipMemcpyDeviceToDevice, st));
    // ---- device -------------------------------------------------------------------------------------------------------
    // the scan in the node map's frame (:190), its NDT map on the node map's lattice (:201-227)
    hipError_t e = ndt_launch_cloud_transform(xyz_dev, count, n_points, stride_bytes, map_stride_bytes, (const double *)(b->dev + L.Tscan),
                                              nullptr, 16, b->xyz_a, st);
    if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "fuser_update: transform", e);
    rc = mapset_build_core(b->scans, first, count, b->xyz_a, n_points, 12, n_points * 12, P.sensor_range, (const double *)(b->dev + L.origin),
                           nullptr, st);
    if (rc != NDTGPU_OK) return rc;
    if (P.discard_cells && n_points > 0) {
        // :229-232 -- ndt_feature::discardCell(ndglobal, cloud.front()) and (.., cloud.back()): the cells of the scan map that hold the
        // first and the last point of the (moved) scan lose their Gaussian
        for (size_t k = 0; k < count; k++) {
            const float *c0 = b->xyz_a + k * n_points * 3;
            e = ndt_launch_discard(b->scans->v, first + k, c0, 1, st);
            if (e == hipSuccess) e = ndt_launch_discard(b->scans->v, first + k, c0 + (n_points - 1) * 3, 1, st);
            if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "fuser_update: discard launch", e);
        }
    }
    // matchFusion / matchFusion2d of the scan map against the slot's node map (:353-357)
    ndtgpu_match_params mp;
    ndtgpu_default_match_params(&mp);
    mp.n_neighbours = P.neighbours; mp.itr_max = P.itr_max; mp.delta_score = P.delta_score; mp.step_control = P.stepcontrol ? 1 : 0;
    mp.dof_mask = P.fusion2d ? 0x23 : 0x3f;
    mp.use_initial_guess = 1;
    NdtMatchParamsDev pd = to_dev(&mp);
    pd.fusion_flags = feat ? (flags | (P.step_control_fusion ? 4 : 0)) : flags;
    const uint32_t *idx_dev = (const uint32_t *)(b->dev + L.idx);
    rc = match_device_core(b->nodes, idx_dev, b->scans, idx_dev, (double *)(b->dev + L.Test), count, pd,
                           (ndtgpu_match_result *)(b->dev + L.match), flags ? (const double *)(b->dev + L.Q) : nullptr, st,
                           feat ? (const unsigned *)(b->dev + L.foff) : nullptr, feat ? (const double *)(b->dev + L.feat) : nullptr);
    if (rc != NDTGPU_OK) return rc;
    // NDTMatcherD2D::covariance at the registered pose (:403-405; a default-constructed matcher: n_neighbours 2)
    if (P.compute_cov) {
        ndtgpu_match_params cp;
        ndtgpu_default_match_params(&cp);
        e = ndt_launch_covariance(b->nodes->v, idx_dev, b->scans->v, idx_dev, (const double *)(b->dev + L.Test), count, cp.n_neighbours,
                                  cp.lfd1, cp.lfd2, P.covariance_mode, (double *)(b->dev + L.cov), (int *)(b->dev + L.covflag), st);
        if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "fuser_update: covariance launch", e);
    }
    // the post-registration step (:361-480), the scan in the frame of the new pose, the fuse-in (:485-486)
    NdtFuserPolicy pol;
    pol.max_translation_norm = P.max_translation_norm; pol.max_rotation_norm = P.max_rotation_norm;
    pol.translation_fuse_delta = 0.05; pol.rotation_fuse_delta = 0.01;                   // ndt_feature_fuser_hmt.h:222-223
    pol.check_consistency = P.check_consistency; pol.fuse_incomplete = P.fuse_incomplete; pol.all_matches_valid = P.all_matches_valid;
    pol.force_odom_as_est = P.force_odom_as_est; pol.compute_cov = P.compute_cov;
    e = ndt_launch_fuser_post(pol, b->sensor_pose_dev, b->st_dev + first, (const double *)(b->dev + L.Tmotion), (const double *)(b->dev + L.Test),
                              (const NdtMatchResultDev *)(b->dev + L.match), P.compute_cov ? (const double *)(b->dev + L.cov) : nullptr,
                              P.compute_cov ? (const int *)(b->dev + L.covflag) : nullptr, count, (double *)(b->dev + L.spose),
                              (double *)(b->dev + L.forigin), (NdtFuserResultDev *)(b->dev + L.res), st);
    if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "fuser_update: post launch", e);
    if (update_ndt_map) {
        e = ndt_launch_cloud_transform(xyz_dev, count, n_points, stride_bytes, map_stride_bytes, (const double *)(b->dev + L.spose), nullptr, 16,
                                       b->xyz_b, st);
        if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "fuser_update: transform", e);
        NdtFuseParams fp;
        fp.maxz = 25.0; fp.sensor_noise = 0.06; fp.maxnumpoints = 1e5; fp.occupancy_limit = 255.0; fp.eval_factor = 1000.0; fp.n_min = 3;   // :485-486
        e = ndt_launch_fuse(b->nodes->v, first, count, b->xyz_b, n_points, 12, n_points * 12, (const double *)(b->dev + L.forigin), fp,
                            b->nodes->nice_range(first, count), st);
        if (e != hipSuccess) return fail(NDTGPU_ERR_HIP, "fuser_update: fuse launch", e);
        { ndtgpu_status trc = b->nodes->touch(st); if (trc != NDTGPU_OK) return trc; }
    }
    HIP_TRY(hipMemcpyAsync(b->pin + L.res, b->dev + L.res, count * sizeof(NdtFuserResultDev), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(b->done_ev, st));
    b->in_flight = true;
    b->fl_first = first; b->fl_count = count;
    b->off_pin_res = L.res;
    b->fl_stream = st;
    return NDTGPU_OK;
}

// the clouds of a *_host entry travel to a device buffer of the bank on a stream of its own; the device entry follows there
static ndtgpu_status fuser_clouds_in(ndtgpu_fuser_bank *b, size_t count, const void *xyz_host, size_t n_points, size_t stride_bytes,
                                     size_t map_stride_bytes, const void **xyz_dev)
{
    *xyz_dev = nullptr;
    if (!count || !n_points) return NDTGPU_OK;
    if (count > 1 && map_stride_bytes < n_points * stride_bytes) return fail(NDTGPU_ERR_INVALID, "fuser: clouds must not overlap");
    ndtgpu_status rc = fuser_catch_up(b);          // (the previous call may still read the buffer)
    if (rc != NDTGPU_OK) return rc;
    if (!b->own_st) HIP_TRY(hipStreamCreateWithFlags(&b->own_st, hipStreamNonBlocking));
    const size_t bytes = (count - 1) * map_stride_bytes + n_points * stride_bytes;
    if (bytes > b->xyz_in_bytes) {
        if (b->xyz_in) (void)hipFree(b->xyz_in);
        b->xyz_in = nullptr; b->xyz_in_bytes = 0;
        HIP_TRY(hipMalloc(&b->xyz_in, bytes));
        b->xyz_in_bytes = bytes;
    }
    HIP_TRY(hipMemcpyAsync(b->xyz_in, xyz_host, bytes, hipMemcpyHostToDevice, b->own_st));
    *xyz_dev = b->xyz_in;
    return NDTGPU_OK;
}

ndtgpu_status ndtgpu_fuser_initialize_batch_host(ndtgpu_fuser_bank *b, size_t first, size_t count, const double *initPose16,
                                                 const void *xyz_host, size_t n_points, size_t stride_bytes, size_t map_stride_bytes)
{
    if (!b || (count && n_points && !xyz_host)) return fail(NDTGPU_ERR_INVALID, "fuser_initialize_host: bad argument");
    const void *dev = nullptr;
    ndtgpu_status rc = fuser_clouds_in(b, count, xyz_host, n_points, stride_bytes, map_stride_bytes, &dev);
    if (rc != NDTGPU_OK) return rc;
    return ndtgpu_fuser_initialize_batch(b, first, count, initPose16, dev, n_points, stride_bytes, map_stride_bytes, (ndtgpu_stream)b->own_st);
}

ndtgpu_status ndtgpu_fuser_update_batch_host(ndtgpu_fuser_bank *b, size_t first, size_t count, const double *Tmotion16,
                                             const void *xyz_host, size_t n_points, size_t stride_bytes, size_t map_stride_bytes,
                                             int update_ndt_map)
{
    if (!b || (count && n_points && !xyz_host)) return fail(NDTGPU_ERR_INVALID, "fuser_update_host: bad argument");
    const void *dev = nullptr;
    ndtgpu_status rc = fuser_clouds_in(b, count, xyz_host, n_points, stride_bytes, map_stride_bytes, &dev);
    if (rc != NDTGPU_OK) return rc;
    return ndtgpu_fuser_update_batch(b, first, count, Tmotion16, dev, n_points, stride_bytes, map_stride_bytes, update_ndt_map,
                                     (ndtgpu_stream)b->own_st);
}

ndtgpu_status ndtgpu_fuser_poses(ndtgpu_fuser_bank *b, size_t first, size_t count, double *Tnow16, ndtgpu_fuser_result *results)
{
    if (!b || first + count > b->n || (count && !Tnow16)) return fail(NDTGPU_ERR_INVALID, "fuser_poses: bad argument");
    const bool had = b->in_flight && b->fl_count > 0;
    const size_t f0 = b->fl_first, fc = b->fl_count, off = b->off_pin_res;
    ndtgpu_status rc = fuser_catch_up(b);
    if (rc != NDTGPU_OK) return rc;
    for (size_t k = 0; k < count; k++) memcpy(Tnow16 + 16 * k, b->st[first + k].s.Tnow, 16 * sizeof(double));
    if (results) {
        // the records of the LAST update call, for the slots it covered (zeroes elsewhere)
        memset(results, 0, count * sizeof *results);
        (void)had;
        if (fc)
            for (size_t k = 0; k < count; k++) {
                const size_t slot = first + k;
                if (slot >= f0 && slot < f0 + fc) memcpy(&results[k], b->pin + off + (slot - f0) * sizeof(NdtFuserResultDev), sizeof(NdtFuserResultDev));
            }
    }
    return NDTGPU_OK;
}

}  // extern "C"
