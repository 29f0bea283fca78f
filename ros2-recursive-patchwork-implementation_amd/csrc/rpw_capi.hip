// rpw_capi.hip -- the C-ABI declared in include/rpw.h: handle, device buffers, streams, launches.
// Host orchestration only; all arithmetic of the path runs in rpw_kernels.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "rpw_internal.h"

using namespace rpw;

namespace {

thread_local std::string g_create_error;

// Streams overlap only when they sit on hardware queues of their own; ROCm reads this knob when the HIP runtime
// starts (default 4). Never overrides a value the user has set.
__attribute__((constructor)) void rpw_runtime_defaults() { (void)setenv("GPU_MAX_HW_QUEUES", "16", 0); }

constexpr int kStageSlots = 8;
constexpr int kSweepRangeMode = 3;   // kSweepRange of rpw_kernels.hip (sweeps over every slot of the node, not over the index list)

struct EventPair { int kernel; hipEvent_t a, b; };

} // namespace

// Everything one batch in flight needs on the device: a handle owns one for the batch / one-frame calls and one
// per pipeline slot (rpw_submit / rpw_wait), so that frames in flight never share scratch memory.
struct Workspace {
    uint32_t max_points = 0, max_frames = 0, max_tiles = 0;
    uint32_t* d_offsets = nullptr;       // [2*(F+1)] frame_off then tile_base, then [F][kMaxSensors+1] sensor offsets
    uint32_t* h_offsets[kStageSlots] = {};   // pinned staging ring
    hipEvent_t stage_ev[kStageSlots] = {};
    bool stage_used[kStageSlots] = {};
    int stage_next = 0;

    uint32_t* d_hist = nullptr;
    uint32_t* d_patch_begin = nullptr;
    PatchDesc* d_patches = nullptr;
    uint32_t* d_ctrl = nullptr;          // [kBuckets] bucket counts + [1] node count (+pad), then [F][PB] class totals of every frame
    uint32_t* d_bucket_list = nullptr;
    float *d_sx = nullptr, *d_sy = nullptr, *d_sz = nullptr, *d_ax = nullptr, *d_ay = nullptr, *d_az = nullptr;
    uint32_t* d_sidx = nullptr;
    unsigned long long* d_mbits = nullptr;   // three mask bit planes
    uint32_t mplane = 0;                    // words per plane
    rpw_frame_stats* d_stats = nullptr;

    // the batch in flight on this workspace: its stream, its size, and an event recorded behind its last kernel
    hipStream_t last_stream = nullptr;
    uint32_t last_frames = 0;
    hipEvent_t done_ev = nullptr;
    bool done_valid = false;
};

// One frame in flight through the host-pointer pipeline (rpw_submit / rpw_wait).
struct Slot {
    Workspace ws;
    hipStream_t stream = nullptr;
    unsigned char* h_in = nullptr;       // pinned, device-readable: the frame as the caller handed it over (read in place by bin_tile)
    size_t h_in_bytes = 0;
    uint8_t* h_labels = nullptr;         // pinned
    uint8_t* d_labels = nullptr;
    float* h_res = nullptr;              // pinned, allocated on first use
    float* d_res = nullptr;
    rpw_frame_stats* h_stats = nullptr;  // pinned
    hipEvent_t done = nullptr;
    bool pending = false;
    bool want_residual = false;
    uint32_t n = 0;
};

struct rpw_handle {
    rpw_config cfg;
    int device = 0;
    uint32_t max_points = 0, max_frames = 0;
    std::string err;

    // config-derived
    int S = 0, P = 0, PB = 0;
    PolarTables tab;
    float* d_bounds = nullptr;

    // LidarFusion pre-step (off when sensors.n == 0)
    SensorTable sensors;

    Workspace ws;                        // batch / one-frame calls
    std::vector<Slot*> slots;            // host-pointer pipeline

    // trace
    int trace = 0;
    uint32_t nodes_cap = 0;
    rpw_node* d_nodes = nullptr;

    // host-pointer path
    hipStream_t own_stream = nullptr;
    unsigned char* h_tail = nullptr;          // pinned: the stats record and the fit error word of a synchronous one-frame call
    float* d_in = nullptr;
    unsigned char* d_raw = nullptr;      // raw PointCloud2 bytes (rpw_segment_pointcloud2)
    size_t raw_bytes = 0;
    uint8_t* d_labels = nullptr;
    float* d_res = nullptr;

    // profiling
    int profiling = 0;
    std::vector<EventPair> pending;
    std::vector<hipEvent_t> free_events;
    double ms[RPW_K_COUNT] = {};
    uint64_t launches[RPW_K_COUNT] = {};
};

namespace {

rpw_status fail(rpw_handle* h, rpw_status st, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return st;
}

#define RPW_HIP(h, call)                                                                          \
    do { hipError_t e_ = (call);                                                                  \
         if (e_ != hipSuccess) return fail((h), RPW_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

void free_workspace(Workspace& w)
{
    auto F = [](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
    F(w.d_offsets); F(w.d_hist); F(w.d_patch_begin); F(w.d_patches);
    F(w.d_ctrl); F(w.d_bucket_list); F(w.d_sx); F(w.d_sy); F(w.d_sz); F(w.d_ax); F(w.d_ay); F(w.d_az);
    F(w.d_sidx); F(w.d_mbits); F(w.d_stats);
    for (int i = 0; i < kStageSlots; ++i) {
        if (w.h_offsets[i]) { (void)hipHostFree(w.h_offsets[i]); w.h_offsets[i] = nullptr; }
        if (w.stage_ev[i]) { (void)hipEventDestroy(w.stage_ev[i]); w.stage_ev[i] = nullptr; }
        w.stage_used[i] = false;
    }
    if (w.done_ev) { (void)hipEventDestroy(w.done_ev); w.done_ev = nullptr; }
    w.done_valid = false; w.last_stream = nullptr; w.last_frames = 0;
}

void free_slot(Slot* s)
{
    if (!s) return;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    free_workspace(s->ws);
    if (s->h_in) (void)hipHostFree(s->h_in);
    if (s->h_labels) (void)hipHostFree(s->h_labels);
    if (s->d_labels) (void)hipFree(s->d_labels);
    if (s->h_res) (void)hipHostFree(s->h_res);
    if (s->d_res) (void)hipFree(s->d_res);
    if (s->h_stats) (void)hipHostFree(s->h_stats);
    if (s->done) (void)hipEventDestroy(s->done);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

void free_device(rpw_handle* h)
{
    auto F = [](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
    for (Slot* s : h->slots) free_slot(s);
    h->slots.clear();
    free_workspace(h->ws);
    F(h->d_bounds); F(h->d_nodes); F(h->d_in); F(h->d_raw); F(h->d_labels); F(h->d_res);
    h->raw_bytes = 0;
}

rpw_status validate_config(rpw_handle* h, const rpw_config* c)
{
    if (!c) return fail(h, RPW_ERR_INVALID, "config is null");
    if (c->num_sectors < 1) return fail(h, RPW_ERR_INVALID, "num_sectors must be >= 1 (got %d)", c->num_sectors);
    if (c->num_sectors > kMaxSectors) return fail(h, RPW_ERR_UNSUPPORTED, "num_sectors %d > %d", c->num_sectors, kMaxSectors);
    if (!(c->filtering_radius == c->filtering_radius)) return fail(h, RPW_ERR_INVALID, "filtering_radius is NaN");
    return RPW_OK;
}

// ring edges and sector bounds with the reference's own expressions (recursive_patchwork.cpp:344-352,364)
void build_tables(const rpw_config& c, PolarTables& t, std::vector<float>& bounds)
{
    const float r_min = 1.0f, r_max = c.filtering_radius;
    const int num_rings = 8;
    for (int i = 0; i <= num_rings; ++i)
        t.edges[i] = r_min * std::pow(r_max / r_min, static_cast<float>(i) / num_rings);
    const float sector_angle = 2.0f * M_PI / c.num_sectors;
    t.radius = c.filtering_radius;
    t.num_sectors = c.num_sectors;
    // thresholds on s = x*x + y*y that are EQUIVALENT to the reference's comparisons of sqrtf(s):
    // sqrtf is correctly rounded and monotone, so { s : sqrtf(s) >= e } is an upper set of floats.
    auto min_s_with_sqrt_ge = [](float e) -> float {
        if (!(e > 0.0f)) return 0.0f;                                   // sqrtf(s) >= e for every s >= 0
        if (std::isinf(e)) return e;
        float s = (float)((double)e * (double)e);
        if (std::isinf(s)) s = std::numeric_limits<float>::max();
        while (s > 0.0f && std::sqrt(std::nextafter(s, 0.0f)) >= e) s = std::nextafter(s, 0.0f);
        while (!(std::sqrt(s) >= e)) { if (s == std::numeric_limits<float>::max()) return std::numeric_limits<float>::infinity(); s = std::nextafter(s, std::numeric_limits<float>::infinity()); }
        return s;
    };
    auto min_s_with_sqrt_gt = [&](float r) -> float {
        if (r < 0.0f) return 0.0f;                                      // nothing is <= a negative radius; s >= 0 always
        if (std::isinf(r)) return r;
        const float up = std::nextafter(r, std::numeric_limits<float>::infinity());
        return min_s_with_sqrt_ge(up);
    };
    t.edges_monotone = 1;
    for (int i = 0; i <= num_rings; ++i) {
        t.ge[i] = min_s_with_sqrt_ge(t.edges[i]);
        if (i && !(t.edges[i] >= t.edges[i - 1])) t.edges_monotone = 0;
    }
    t.gt_radius = min_s_with_sqrt_gt(c.filtering_radius);
    t.inv_sector = 1.0f / sector_angle;
    {   // cheap angle in sector units (approximate path: plain float arithmetic is fine here)
        const float coef[5] = {0.9998660f, -0.3302995f, 0.1801410f, -0.0851330f, 0.0208351f};
        for (int i = 0; i < 5; ++i) t.ac[i] = coef[i] * t.inv_sector;
        t.uq = 0.25f * (float)c.num_sectors; t.uh = 0.5f * (float)c.num_sectors; t.uf = (float)c.num_sectors;
        t.usure = 0.5f - 1.0e-4f * t.inv_sector;
    }
    bounds.resize((size_t)c.num_sectors + 1);
    for (int k = 0; k <= c.num_sectors; ++k) bounds[k] = k * sector_angle;
}

// device buffers of one workspace: N points / F frames of capacity
rpw_status alloc_workspace(rpw_handle* h, Workspace& w, size_t N, size_t F)
{
    free_workspace(w);
    if (N == 0) N = 1;
    if (F == 0) F = 1;
    w.max_points = (uint32_t)N; w.max_frames = (uint32_t)F;
    const size_t off_words = 2 * (F + 1) + F * (kMaxSensors + 1);
    RPW_HIP(h, hipMalloc(&w.d_offsets, off_words * sizeof(uint32_t)));
    for (int i = 0; i < kStageSlots; ++i) {
        RPW_HIP(h, hipHostMalloc((void**)&w.h_offsets[i], off_words * sizeof(uint32_t), hipHostMallocDefault));
        RPW_HIP(h, hipEventCreateWithFlags(&w.stage_ev[i], hipEventDisableTiming));
    }
    RPW_HIP(h, hipEventCreateWithFlags(&w.done_ev, hipEventDisableTiming));
    w.max_tiles = (uint32_t)(N / kTilePoints + F);
    RPW_HIP(h, hipMalloc(&w.d_hist, (size_t)w.max_tiles * h->PB * sizeof(uint32_t)));
    RPW_HIP(h, hipMalloc(&w.d_patch_begin, F * (size_t)(h->P + 1) * sizeof(uint32_t)));
    RPW_HIP(h, hipMalloc(&w.d_patches, F * (size_t)h->P * sizeof(PatchDesc)));
    RPW_HIP(h, hipMalloc(&w.d_ctrl, (kBuckets + 4 + F * (size_t)h->PB) * sizeof(uint32_t)));     // + per-frame class totals
    RPW_HIP(h, hipMalloc(&w.d_bucket_list, (size_t)kBuckets * F * h->P * sizeof(uint32_t)));
    RPW_HIP(h, hipMalloc(&w.d_sx, N * sizeof(float)));
    RPW_HIP(h, hipMalloc(&w.d_sy, N * sizeof(float)));
    RPW_HIP(h, hipMalloc(&w.d_sz, N * sizeof(float)));
    RPW_HIP(h, hipMalloc(&w.d_ax, N * sizeof(float)));
    RPW_HIP(h, hipMalloc(&w.d_ay, N * sizeof(float)));
    RPW_HIP(h, hipMalloc(&w.d_az, N * sizeof(float)));
    RPW_HIP(h, hipMalloc(&w.d_sidx, N * sizeof(uint32_t)));
    w.mplane = (uint32_t)(N / 64 + F * (size_t)h->P + 8);      // a patch starts at word (begin >> 6) + its index in the batch
    RPW_HIP(h, hipMalloc(&w.d_mbits, 3 * (size_t)w.mplane * sizeof(unsigned long long)));
    RPW_HIP(h, hipMalloc(&w.d_stats, F * sizeof(rpw_frame_stats)));
    return RPW_OK;
}

rpw_status alloc_slot(rpw_handle* h, Slot* s)
{
    const size_t N = h->max_points ? h->max_points : 1;
    rpw_status st = alloc_workspace(h, s->ws, N, 1);
    if (st != RPW_OK) return st;
    RPW_HIP(h, hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    RPW_HIP(h, hipEventCreateWithFlags(&s->done, hipEventDisableTiming));
    s->h_in_bytes = N * 3 * sizeof(float);
    RPW_HIP(h, hipHostMalloc((void**)&s->h_in, s->h_in_bytes, hipHostMallocDefault));
    RPW_HIP(h, hipHostMalloc((void**)&s->h_labels, N, hipHostMallocDefault));
    RPW_HIP(h, hipMalloc((void**)&s->d_labels, N));
    RPW_HIP(h, hipHostMalloc((void**)&s->h_stats, sizeof(rpw_frame_stats) + 16, hipHostMallocDefault));
    return RPW_OK;
}

rpw_status allocate(rpw_handle* h)
{
    const size_t n_slots = h->slots.size();
    free_device(h);
    const size_t N = h->max_points ? h->max_points : 1, F = h->max_frames ? h->max_frames : 1;
    h->S = h->cfg.num_sectors; h->P = 8 * h->S; h->PB = (h->P + 4 + 3) & ~3;       // patch classes + {no patch, outside, non-finite, ego-removed}
    std::vector<float> bounds;
    build_tables(h->cfg, h->tab, bounds);
    RPW_HIP(h, hipMalloc(&h->d_bounds, bounds.size() * sizeof(float)));
    RPW_HIP(h, hipMemcpy(h->d_bounds, bounds.data(), bounds.size() * sizeof(float), hipMemcpyHostToDevice));
    rpw_status st = alloc_workspace(h, h->ws, N, F);
    if (st != RPW_OK) return st;
    if (h->trace && h->nodes_cap) RPW_HIP(h, hipMalloc(&h->d_nodes, (size_t)h->nodes_cap * sizeof(rpw_node)));
    for (size_t i = 0; i < n_slots; ++i) {           // a pipeline that existed before a re-allocation is rebuilt
        Slot* s = new (std::nothrow) Slot();
        if (!s) return fail(h, RPW_ERR_INVALID, "out of host memory");
        h->slots.push_back(s);
        st = alloc_slot(h, s);
        if (st != RPW_OK) return st;
    }
    return RPW_OK;
}

hipEvent_t get_event(rpw_handle* h)
{
    if (!h->free_events.empty()) { hipEvent_t e = h->free_events.back(); h->free_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void drain_profile(rpw_handle* h)
{
    for (EventPair& ep : h->pending) {
        (void)hipEventSynchronize(ep.b);
        float t = 0.0f;
        if (hipEventElapsedTime(&t, ep.a, ep.b) == hipSuccess) { h->ms[ep.kernel] += (double)t; h->launches[ep.kernel]++; }
        h->free_events.push_back(ep.a); h->free_events.push_back(ep.b);
    }
    h->pending.clear();
}

template <class Launch>
void timed(rpw_handle* h, int kernel, hipStream_t s, Launch launch)
{
    if (!h->profiling) { launch(); return; }
    EventPair ep; ep.kernel = kernel; ep.a = get_event(h); ep.b = get_event(h);
    if (!ep.a || !ep.b) {                        // no event to be had: run the kernel untimed rather than not at all
        if (ep.a) h->free_events.push_back(ep.a);
        if (ep.b) h->free_events.push_back(ep.b);
        launch(); return;
    }
    (void)hipEventRecord(ep.a, s);
    launch();
    (void)hipEventRecord(ep.b, s);
    h->pending.push_back(ep);
    if (h->pending.size() > 8192) drain_profile(h);
}

struct InputLayout { uint32_t stride = 12, off[3] = {0, 4, 8}; };

// kernel parameters of one batch on workspace w
void fill_params(rpw_handle* h, Workspace& w, DevParams& p, const void* d_xyz, uint32_t F, uint32_t tiles, uint8_t* d_labels,
                 float* d_residual, const InputLayout& lay, int ns)
{
    p.cfg = h->cfg; p.tab = h->tab; p.bounds = h->d_bounds; p.S = h->S; p.P = h->P; p.PB = h->PB;
    p.n_frames = F; p.n_tiles = tiles; p.frame_first = 0; p.frame_count = F; p.tile_first = 0; p.tile_count = tiles; p.frame_off = w.d_offsets; p.tile_base = w.d_offsets + (F + 1);
    p.xyz = reinterpret_cast<const P3*>(d_xyz); p.in_stride = lay.stride;
    p.in_off[0] = lay.off[0]; p.in_off[1] = lay.off[1]; p.in_off[2] = lay.off[2]; p.labels = d_labels;
    p.sensors = h->sensors; if (ns == 0) p.sensors.n = 0;
    p.sensor_off = w.d_offsets + 2 * (F + 1); p.residual = d_residual;
    p.hist = w.d_hist; p.ftot = w.d_ctrl + kBuckets + 4; p.patch_begin = w.d_patch_begin; p.patches = w.d_patches;
    p.bucket_count = w.d_ctrl; p.bucket_list = w.d_bucket_list; p.list_stride = w.max_frames * (uint32_t)h->P;
    p.sx = w.d_sx; p.sy = w.d_sy; p.sz = w.d_sz; p.sidx = w.d_sidx; p.ax = w.d_ax; p.ay = w.d_ay; p.az = w.d_az;
    p.mbits = w.d_mbits; p.mplane = w.mplane; p.stats = w.d_stats;
    { static const int remap = [] { const char* e = getenv("RPW_BIN_REMAP"); return e ? atoi(e) : 1; }(); p.bin_remap = remap; }
    // the trace buffer belongs to the handle: only its own workspace writes it
    const bool tracing = (&w == &h->ws) && h->trace && h->d_nodes;
    p.trace = tracing ? 1 : 0; p.nodes = tracing ? h->d_nodes : nullptr; p.node_count = w.d_ctrl + kBuckets; p.work = w.d_ctrl + kBuckets + 1; p.nodes_cap = tracing ? h->nodes_cap : 0;
}

rpw_status run_batch(rpw_handle* h, Workspace& w, const void* d_xyz, const uint32_t* frame_offsets, uint32_t n_frames,
                     uint8_t* d_labels, float* d_residual, hipStream_t stream, const InputLayout& lay = InputLayout(),
                     const uint32_t* sensor_offsets = nullptr /* [F][sensors.n + 1], frame-relative */)
{
    if (!frame_offsets) return fail(h, RPW_ERR_INVALID, "frame_offsets is null");
    if (n_frames > w.max_frames) return fail(h, RPW_ERR_CAPACITY, "n_frames %u > capacity %u", n_frames, w.max_frames);
    if (n_frames == 0) {
        // an empty batch touches nothing: it completes at once, whatever ran before
        w.last_stream = stream; w.last_frames = 0; w.done_valid = false;
        return RPW_OK;
    }
    const uint32_t total = frame_offsets[n_frames];
    if (total > w.max_points) return fail(h, RPW_ERR_CAPACITY, "batch of %u points > capacity %u", total, w.max_points);
    if (total && (!d_xyz || !d_labels)) return fail(h, RPW_ERR_INVALID, "null device pointer");
    RPW_HIP(h, hipSetDevice(h->device));

    // ---- validate everything on the host before anything is enqueued ------------------------------------------
    const uint32_t F = n_frames;
    if (frame_offsets[0] != 0) return fail(h, RPW_ERR_INVALID, "frame_offsets[0] must be 0");
    uint32_t tiles = 0;
    for (uint32_t f = 0; f < F; ++f) {
        if (frame_offsets[f + 1] < frame_offsets[f]) return fail(h, RPW_ERR_INVALID, "frame_offsets not monotone");
        const uint32_t nf = frame_offsets[f + 1] - frame_offsets[f];
        if ((uint64_t)nf * 16u > 0xffffffffull)      // bin_tile addresses a frame's records with 32-bit byte offsets
            return fail(h, RPW_ERR_CAPACITY, "frame %u has %u points; the limit is 268435455 per frame", f, nf);
        tiles += (nf + kTilePoints - 1) / kTilePoints;
    }
    if (tiles > w.max_tiles) return fail(h, RPW_ERR_CAPACITY, "tile count %u > capacity %u", tiles, w.max_tiles);
    const int ns = sensor_offsets ? h->sensors.n : 0;
    for (uint32_t f = 0; ns > 0 && f < F; ++f) {
        const uint32_t nf = frame_offsets[f + 1] - frame_offsets[f];
        const uint32_t* src = sensor_offsets + (size_t)f * (ns + 1);
        if (src[0] != 0 || src[ns] != nf) return fail(h, RPW_ERR_INVALID, "sensor offsets of frame %u must run from 0 to the frame size", f);
        for (int j = 1; j <= ns; ++j)
            if (src[j] < src[j - 1]) return fail(h, RPW_ERR_INVALID, "sensor offsets of frame %u not monotone", f);
    }

    // ---- the workspace is single-occupancy: a batch on ANOTHER stream first waits for the one in flight ------------
    // (scratch buffers, control words and stats belong to the workspace; same-stream reuse is ordered by the stream)
    if (w.done_valid && w.last_stream != stream) RPW_HIP(h, hipStreamWaitEvent(stream, w.done_ev, 0));

    const int slot = w.stage_next; w.stage_next = (w.stage_next + 1) % kStageSlots;
    if (w.stage_used[slot]) RPW_HIP(h, hipEventSynchronize(w.stage_ev[slot]));
    uint32_t* ho = w.h_offsets[slot];
    tiles = 0;
    for (uint32_t f = 0; f <= F; ++f) {
        ho[f] = frame_offsets[f];
        ho[(F + 1) + f] = tiles;
        if (f < F) tiles += (frame_offsets[f + 1] - frame_offsets[f] + kTilePoints - 1) / kTilePoints;
    }
    size_t copy_words = 2 * (size_t)(F + 1);
    if (ns > 0) {
        uint32_t* so = ho + 2 * (F + 1);
        std::memcpy(so, sensor_offsets, (size_t)F * (ns + 1) * sizeof(uint32_t));
        copy_words += (size_t)F * (ns + 1);
    }
    RPW_HIP(h, hipMemcpyAsync(w.d_offsets, ho, copy_words * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    RPW_HIP(h, hipEventRecord(w.stage_ev[slot], stream));
    w.stage_used[slot] = true;
    RPW_HIP(h, hipMemsetAsync(w.d_ctrl, 0, (kBuckets + 4 + (size_t)F * h->PB) * sizeof(uint32_t), stream));

    DevParams p;
    fill_params(h, w, p, d_xyz, F, tiles, d_labels, d_residual, lay, ns);

    // ---- binning pass (count, scan, scatter: a stable counting sort by patch) once over the whole batch, then the fit ----
    timed(h, RPW_K_BIN_COUNT, stream, [&] { launch_bin_count(p, stream); });
    timed(h, RPW_K_SCAN, stream, [&] { launch_scan(p, stream); });
    timed(h, RPW_K_BIN_SCATTER, stream, [&] { launch_bin_scatter(p, stream); });
    timed(h, RPW_K_FIT, stream, [&] { launch_fit(p, stream); });
    RPW_HIP(h, hipGetLastError());
    RPW_HIP(h, hipEventRecord(w.done_ev, stream));
    w.done_valid = true; w.last_stream = stream; w.last_frames = n_frames;     // only a batch that was enqueued becomes "the last batch"
    return RPW_OK;
}

// Layout check shared by every PointCloud2 entry: FLOAT32 x / y / z fields, 4-byte aligned, inside the record.
// Written without additions that could wrap (off_x = 0xFFFFFFFC must not pass).
bool pc2_layout_ok(uint32_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z)
{
    if (point_step < 12 || (point_step & 3u)) return false;
    const uint32_t last = point_step - 4;
    return !(off_x & 3u) && !(off_y & 3u) && !(off_z & 3u) && off_x <= last && off_y <= last && off_z <= last;
}

Slot* get_slot(rpw_handle* h, uint32_t slot)
{
    return slot < h->slots.size() ? h->slots[slot] : nullptr;
}

// common tail of rpw_submit / rpw_submit_pointcloud2. The slot's stream receives KERNELS ONLY: the frame is read in
// place from the slot's pinned buffer by the binning kernel (12 bytes per point over the host link, once) and the labels,
// residuals and stats are written to the slot's pinned result buffers by a copy kernel. Copy-engine operations between
// the kernels of different slots made the slots wait for each other (8 slots ran little faster than 2).
rpw_status enqueue_slot(rpw_handle* h, Slot* s, uint32_t n, size_t bytes, const InputLayout& lay, bool want_residual)
{
    (void)bytes;
    const size_t N = h->max_points ? h->max_points : 1;
    if (want_residual && !s->d_res) {
        RPW_HIP(h, hipMalloc((void**)&s->d_res, N * sizeof(float)));
        RPW_HIP(h, hipHostMalloc((void**)&s->h_res, N * sizeof(float), hipHostMallocDefault));
    }
    hipStream_t st = s->stream;
    Workspace& w = s->ws;
    if ((uint64_t)n * 16u > 0xffffffffull) return fail(h, RPW_ERR_CAPACITY, "frame of %u points; the limit is 268435455 per frame", n);
    const uint32_t tiles = (n + kTilePoints - 1) / kTilePoints;
    if (tiles > w.max_tiles) return fail(h, RPW_ERR_CAPACITY, "tile count %u > capacity %u", tiles, w.max_tiles);
    launch_frame_setup(w.d_offsets, n, tiles, w.d_ctrl, (uint32_t)(kBuckets + 4 + h->PB), st);
    DevParams p;
    fill_params(h, w, p, s->h_in, 1, tiles, s->d_labels, want_residual ? s->d_res : nullptr, lay, 0);
    p.trace = 0; p.nodes = nullptr; p.nodes_cap = 0;
    timed(h, RPW_K_BIN_COUNT, st, [&] { launch_bin_count(p, st); });
    timed(h, RPW_K_SCAN, st, [&] { launch_scan(p, st); });
    timed(h, RPW_K_BIN_SCATTER, st, [&] { launch_bin_scatter(p, st); });
    timed(h, RPW_K_FIT, st, [&] { launch_fit(p, st); });
    launch_frame_copy_out(s->d_labels, s->h_labels, n, want_residual ? s->d_res : nullptr, s->h_res, w.d_stats, s->h_stats, w.d_ctrl + kBuckets + 2, st);
    RPW_HIP(h, hipGetLastError());
    RPW_HIP(h, hipEventRecord(s->done, st));
    s->pending = true; s->want_residual = want_residual; s->n = n;
    return RPW_OK;
}

} // namespace

extern "C" {

uint32_t rpw_abi_version(void) { return RPW_ABI_VERSION; }

void rpw_default_config(rpw_config* c)
{
    if (!c) return;
    c->sensor_height = 1.2f; c->max_range = 150.0f; c->num_sectors = 10; c->max_iter = 100;
    c->adaptive_seed_height = 1; c->th_seeds = 0.15f; c->th_dist = 0.2f; c->th_outlier = 0.08f;
    c->filtering_radius = 150.0f; c->max_split_depth = 1000;
}

const char* rpw_last_error(const rpw_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

rpw_status rpw_create(const rpw_config* cfg, int device, uint32_t max_points, uint32_t max_frames, rpw_handle** out)
{
    if (!out) return fail(nullptr, RPW_ERR_INVALID, "out is null");
    *out = nullptr;
    rpw_config c;
    if (cfg) c = *cfg; else rpw_default_config(&c);
    rpw_status st = validate_config(nullptr, &c);
    if (st != RPW_OK) return st;
    if (max_points > (1u << 30))      // patch_fit addresses its per-slot arrays with 32-bit byte offsets
        return fail(nullptr, RPW_ERR_CAPACITY, "max_points %u > 1073741824", max_points);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, RPW_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= count) return fail(nullptr, RPW_ERR_NO_DEVICE, "device %d out of range (have %d)", device, count);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, RPW_ERR_NO_DEVICE, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(nullptr, RPW_ERR_NO_DEVICE, "device %d is %s; kernels are built for gfx950 only", device, prop.gcnArchName);
    }
    rpw_handle* h = new (std::nothrow) rpw_handle();
    if (!h) return fail(nullptr, RPW_ERR_INVALID, "out of host memory");
    std::memset(&h->sensors, 0, sizeof h->sensors);
    h->cfg = c; h->device = device; h->max_points = max_points; h->max_frames = max_frames ? max_frames : 1;
    // The handle's own stream is created lazily (by the host-pointer path): ROCm multiplexes all streams of a
    // process onto a few hardware queues, and an idle extra stream per handle is enough to make the caller's
    // streams share queues and serialise.
    st = allocate(h);
    if (st != RPW_OK) { g_create_error = h->err; rpw_destroy(h); return st; }
    *out = h;
    return RPW_OK;
}

void rpw_destroy(rpw_handle* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    drain_profile(h);
    for (hipEvent_t e : h->free_events) (void)hipEventDestroy(e);
    free_device(h);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    if (h->h_tail) (void)hipHostFree(h->h_tail);
    delete h;
}

rpw_status rpw_set_config(rpw_handle* h, const rpw_config* cfg)
{
    if (!h) return RPW_ERR_INVALID;
    rpw_status st = validate_config(h, cfg);
    if (st != RPW_OK) return st;
    RPW_HIP(h, hipSetDevice(h->device));
    RPW_HIP(h, hipDeviceSynchronize());
    const bool realloc = cfg->num_sectors != h->cfg.num_sectors;
    h->cfg = *cfg;
    if (realloc) return allocate(h);
    std::vector<float> bounds;
    build_tables(h->cfg, h->tab, bounds);
    RPW_HIP(h, hipMemcpy(h->d_bounds, bounds.data(), bounds.size() * sizeof(float), hipMemcpyHostToDevice));
    return RPW_OK;
}

rpw_status rpw_get_config(const rpw_handle* h, rpw_config* cfg)
{
    if (!h || !cfg) return RPW_ERR_INVALID;
    *cfg = h->cfg;
    return RPW_OK;
}

rpw_status rpw_segment_batch_device(rpw_handle* h, const float* d_xyz, const uint32_t* frame_offsets, uint32_t n_frames,
                                    uint8_t* d_labels, float* d_residual, void* stream)
{
    if (!h) return RPW_ERR_INVALID;
    return run_batch(h, h->ws, d_xyz, frame_offsets, n_frames, d_labels, d_residual, (hipStream_t)stream);
}

// patch_fit's chain-server form hands work between waves through LDS words and polls with a spin limit: a protocol error
// ends the launch with this flag set instead of hanging the GPU. The batch's stream must have completed.
static const char* const kFitErrorText = "patch_fit: a wave timed out waiting for the chain server (results of this batch are invalid)";
static rpw_status check_fit_error(rpw_handle* h, Workspace& w)
{
    uint32_t flag = 0;
    RPW_HIP(h, hipMemcpy(&flag, w.d_ctrl + kBuckets + 2, sizeof flag, hipMemcpyDeviceToHost));
    return flag ? fail(h, RPW_ERR_HIP, "%s", kFitErrorText) : RPW_OK;
}

// The tail of a synchronous one-frame call: the frame's stats record and the fit error word come back on the call's own
// stream, into pinned memory, in front of the one synchronisation (two blocking 4- and 64-byte copies behind it cost ~40 us).
static rpw_status finish_one_frame(rpw_handle* h, hipStream_t s, rpw_frame_stats* stats)
{
    if (!h->h_tail) RPW_HIP(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_tail), sizeof(rpw_frame_stats) + 16));
    RPW_HIP(h, hipMemcpyAsync(h->h_tail, h->ws.d_stats, sizeof(rpw_frame_stats), hipMemcpyDeviceToHost, s));
    RPW_HIP(h, hipMemcpyAsync(h->h_tail + sizeof(rpw_frame_stats), h->ws.d_ctrl + kBuckets + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    RPW_HIP(h, hipStreamSynchronize(s));
    if (stats) memcpy(stats, h->h_tail, sizeof(rpw_frame_stats));
    uint32_t flag = 0;
    memcpy(&flag, h->h_tail + sizeof(rpw_frame_stats), sizeof flag);
    return flag ? fail(h, RPW_ERR_HIP, "%s", kFitErrorText) : RPW_OK;
}

rpw_status rpw_get_batch_stats(rpw_handle* h, rpw_frame_stats* stats, uint32_t n_frames)
{
    if (!h || !stats) return RPW_ERR_INVALID;
    if (n_frames > h->ws.last_frames) return fail(h, RPW_ERR_INVALID, "asked for %u frames, last batch had %u", n_frames, h->ws.last_frames);
    RPW_HIP(h, hipSetDevice(h->device));
    RPW_HIP(h, hipStreamSynchronize(h->ws.last_stream));
    if (n_frames) RPW_HIP(h, hipMemcpy(stats, h->ws.d_stats, n_frames * sizeof(rpw_frame_stats), hipMemcpyDeviceToHost));
    return check_fit_error(h, h->ws);
}

rpw_status rpw_result_record_device(rpw_handle* h, int32_t* d_out, uint32_t n_frames, void* stream)
{
    if (!h || (n_frames && !d_out)) return RPW_ERR_INVALID;
    Workspace& w = h->ws;
    if (n_frames > w.last_frames) return fail(h, RPW_ERR_INVALID, "asked for %u frames, last batch had %u", n_frames, w.last_frames);
    RPW_HIP(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if (w.done_valid && s != w.last_stream) RPW_HIP(h, hipStreamWaitEvent(s, w.done_ev, 0));   // behind the batch, whatever the stream
    launch_result_record(w.d_stats, d_out, n_frames, w.d_ctrl + kBuckets + 2, s);      // (+ patch_fit's error flag: records of an invalid batch are -1)
    RPW_HIP(h, hipGetLastError());
    return RPW_OK;
}

rpw_status rpw_segment(rpw_handle* h, const float* xyz, uint32_t n, uint8_t* labels, float* residual, rpw_frame_stats* stats)
{
    if (!h) return RPW_ERR_INVALID;
    if (n && (!xyz || !labels)) return fail(h, RPW_ERR_INVALID, "null pointer");
    if (n > h->max_points) return fail(h, RPW_ERR_CAPACITY, "frame of %u points > capacity %u", n, h->max_points);
    RPW_HIP(h, hipSetDevice(h->device));
    const size_t N = h->max_points ? h->max_points : 1;
    if (!h->d_in) RPW_HIP(h, hipMalloc(&h->d_in, N * 3 * sizeof(float)));
    if (!h->d_labels) RPW_HIP(h, hipMalloc(&h->d_labels, N));
    if (residual && !h->d_res) RPW_HIP(h, hipMalloc(&h->d_res, N * sizeof(float)));
    if (!h->own_stream) RPW_HIP(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    hipStream_t s = h->own_stream;
    if (n) RPW_HIP(h, hipMemcpyAsync(h->d_in, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, s));
    const uint32_t off[2] = {0, n};
    rpw_status st = run_batch(h, h->ws, h->d_in, off, 1, h->d_labels, residual ? h->d_res : nullptr, s);
    if (st != RPW_OK) return st;
    if (n) RPW_HIP(h, hipMemcpyAsync(labels, h->d_labels, n, hipMemcpyDeviceToHost, s));
    if (n && residual) RPW_HIP(h, hipMemcpyAsync(residual, h->d_res, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s));
    return finish_one_frame(h, s, stats);
}

rpw_status rpw_segment_pointcloud2(rpw_handle* h, const uint8_t* data, uint32_t n, uint32_t point_step,
                                   uint32_t off_x, uint32_t off_y, uint32_t off_z,
                                   uint8_t* labels, float* residual, rpw_frame_stats* stats)
{
    if (!h) return RPW_ERR_INVALID;
    if (n && (!data || !labels)) return fail(h, RPW_ERR_INVALID, "null pointer");
    if (!pc2_layout_ok(point_step, off_x, off_y, off_z))
        return fail(h, RPW_ERR_INVALID, "PointCloud2 layout: point_step %u, x/y/z offsets %u/%u/%u must be 4-byte aligned FLOAT32 fields inside the record",
                    point_step, off_x, off_y, off_z);
    if (n > h->max_points) return fail(h, RPW_ERR_CAPACITY, "frame of %u points > capacity %u", n, h->max_points);
    RPW_HIP(h, hipSetDevice(h->device));
    const size_t N = h->max_points ? h->max_points : 1;
    const size_t need = (size_t)n * point_step;
    if (need > h->raw_bytes) {                       // staging for the raw message bytes, grown on demand
        if (h->d_raw) { (void)hipFree(h->d_raw); h->d_raw = nullptr; h->raw_bytes = 0; }
        RPW_HIP(h, hipMalloc(&h->d_raw, need));
        h->raw_bytes = need;
    }
    if (!h->d_labels) RPW_HIP(h, hipMalloc(&h->d_labels, N));
    if (residual && !h->d_res) RPW_HIP(h, hipMalloc(&h->d_res, N * sizeof(float)));
    if (!h->own_stream) RPW_HIP(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    hipStream_t s = h->own_stream;
    if (n) RPW_HIP(h, hipMemcpyAsync(h->d_raw, data, need, hipMemcpyHostToDevice, s));
    const uint32_t off[2] = {0, n};
    InputLayout lay; lay.stride = point_step; lay.off[0] = off_x; lay.off[1] = off_y; lay.off[2] = off_z;
    rpw_status st = run_batch(h, h->ws, h->d_raw, off, 1, h->d_labels, residual ? h->d_res : nullptr, s, lay);
    if (st != RPW_OK) return st;
    if (n) RPW_HIP(h, hipMemcpyAsync(labels, h->d_labels, n, hipMemcpyDeviceToHost, s));
    if (n && residual) RPW_HIP(h, hipMemcpyAsync(residual, h->d_res, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s));
    return finish_one_frame(h, s, stats);
}

/* ---- host-pointer pipeline: K frames in flight, one slot each ---------------------------------------------------- */

rpw_status rpw_set_pipeline(rpw_handle* h, uint32_t n_slots)
{
    if (!h) return RPW_ERR_INVALID;
    if (n_slots > RPW_MAX_SLOTS) return fail(h, RPW_ERR_UNSUPPORTED, "%u pipeline slots > %d", n_slots, RPW_MAX_SLOTS);
    RPW_HIP(h, hipSetDevice(h->device));
    while (h->slots.size() > n_slots) { free_slot(h->slots.back()); h->slots.pop_back(); }   // (waits for the slot's frame)
    while (h->slots.size() < n_slots) {
        Slot* s = new (std::nothrow) Slot();
        if (!s) return fail(h, RPW_ERR_INVALID, "out of host memory");
        h->slots.push_back(s);
        rpw_status st = alloc_slot(h, s);
        if (st != RPW_OK) { free_slot(s); h->slots.pop_back(); return st; }
    }
    return RPW_OK;
}

uint32_t rpw_get_pipeline(const rpw_handle* h) { return h ? (uint32_t)h->slots.size() : 0u; }

float* rpw_slot_input(rpw_handle* h, uint32_t slot)
{
    Slot* s = h ? get_slot(h, slot) : nullptr;
    return s ? reinterpret_cast<float*>(s->h_in) : nullptr;
}

rpw_status rpw_submit(rpw_handle* h, uint32_t slot, const float* xyz, uint32_t n, uint32_t flags)
{
    if (!h) return RPW_ERR_INVALID;
    Slot* s = get_slot(h, slot);
    if (!s) return fail(h, RPW_ERR_INVALID, "slot %u does not exist (rpw_set_pipeline made %zu)", slot, h->slots.size());
    if (s->pending) return fail(h, RPW_ERR_BUSY, "slot %u still holds a frame: rpw_wait it first", slot);
    if (n && !xyz) return fail(h, RPW_ERR_INVALID, "null pointer");
    if (n > h->max_points) return fail(h, RPW_ERR_CAPACITY, "frame of %u points > capacity %u", n, h->max_points);
    RPW_HIP(h, hipSetDevice(h->device));
    const size_t bytes = (size_t)n * 3 * sizeof(float);
    if (!s->h_in || bytes > s->h_in_bytes) return fail(h, RPW_ERR_HIP, "slot %u has no staging buffer for %zu bytes", slot, bytes);
    if (n && reinterpret_cast<const unsigned char*>(xyz) != s->h_in) std::memcpy(s->h_in, xyz, bytes);   // filled in place: nothing to copy
    return enqueue_slot(h, s, n, bytes, InputLayout(), (flags & RPW_SUBMIT_RESIDUAL) != 0);
}

rpw_status rpw_submit_pointcloud2(rpw_handle* h, uint32_t slot, const uint8_t* data, uint32_t n, uint32_t point_step,
                                  uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t flags)
{
    if (!h) return RPW_ERR_INVALID;
    Slot* s = get_slot(h, slot);
    if (!s) return fail(h, RPW_ERR_INVALID, "slot %u does not exist (rpw_set_pipeline made %zu)", slot, h->slots.size());
    if (s->pending) return fail(h, RPW_ERR_BUSY, "slot %u still holds a frame: rpw_wait it first", slot);
    if (n && !data) return fail(h, RPW_ERR_INVALID, "null pointer");
    if (!pc2_layout_ok(point_step, off_x, off_y, off_z))
        return fail(h, RPW_ERR_INVALID, "PointCloud2 layout: point_step %u, x/y/z offsets %u/%u/%u must be 4-byte aligned FLOAT32 fields inside the record",
                    point_step, off_x, off_y, off_z);
    if (n > h->max_points) return fail(h, RPW_ERR_CAPACITY, "frame of %u points > capacity %u", n, h->max_points);
    RPW_HIP(h, hipSetDevice(h->device));
    const size_t bytes = (size_t)n * point_step;
    if (bytes > s->h_in_bytes) {
        // records wider than 12 bytes: the slot's staging grows on demand. The new buffer is allocated FIRST and swapped in
        // only on success, so a failed allocation leaves the slot usable (pointers from rpw_slot_input die with the old one)
        unsigned char* grown = nullptr;
        RPW_HIP(h, hipHostMalloc((void**)&grown, bytes, hipHostMallocDefault));
        const bool in_place = data == s->h_in;        // the caller filled the old buffer: its bytes move along
        if (in_place && s->h_in_bytes) std::memcpy(grown, s->h_in, s->h_in_bytes);
        if (s->h_in) (void)hipHostFree(s->h_in);
        s->h_in = grown; s->h_in_bytes = bytes;
        if (in_place) data = grown;
    }
    if (!s->h_in) return fail(h, RPW_ERR_HIP, "slot %u has no staging buffer", slot);
    if (bytes && data != s->h_in) std::memcpy(s->h_in, data, bytes);
    InputLayout lay; lay.stride = point_step; lay.off[0] = off_x; lay.off[1] = off_y; lay.off[2] = off_z;
    return enqueue_slot(h, s, n, bytes, lay, (flags & RPW_SUBMIT_RESIDUAL) != 0);
}

rpw_status rpw_poll(rpw_handle* h, uint32_t slot, int* done)
{
    if (!h || !done) return RPW_ERR_INVALID;
    Slot* s = get_slot(h, slot);
    if (!s) return fail(h, RPW_ERR_INVALID, "slot %u does not exist", slot);
    if (!s->pending) return fail(h, RPW_ERR_INVALID, "slot %u holds no frame", slot);
    const hipError_t e = hipEventQuery(s->done);
    if (e == hipSuccess) { *done = 1; return RPW_OK; }
    if (e == hipErrorNotReady) { *done = 0; return RPW_OK; }
    return fail(h, RPW_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
}

rpw_status rpw_wait(rpw_handle* h, uint32_t slot, uint8_t* labels, float* residual, rpw_frame_stats* stats)
{
    if (!h) return RPW_ERR_INVALID;
    Slot* s = get_slot(h, slot);
    if (!s) return fail(h, RPW_ERR_INVALID, "slot %u does not exist", slot);
    if (!s->pending) return fail(h, RPW_ERR_INVALID, "slot %u holds no frame: rpw_submit first", slot);
    if (s->n && !labels) return fail(h, RPW_ERR_INVALID, "labels is null");
    if (residual && !s->want_residual) return fail(h, RPW_ERR_INVALID, "slot %u was submitted without RPW_SUBMIT_RESIDUAL", slot);
    RPW_HIP(h, hipEventSynchronize(s->done));
    s->pending = false;                               // the frame is done, whatever the caller takes of it
    if (s->n) std::memcpy(labels, s->h_labels, s->n);
    if (s->n && residual) std::memcpy(residual, s->h_res, (size_t)s->n * sizeof(float));
    if (stats) *stats = *s->h_stats;
    if (*reinterpret_cast<const uint32_t*>(s->h_stats + 1)) return fail(h, RPW_ERR_HIP, "%s", kFitErrorText);
    return RPW_OK;
}

rpw_status rpw_set_sensors(rpw_handle* h, uint32_t n_sensors, const rpw_sensor* sensors)
{
    if (!h) return RPW_ERR_INVALID;
    if (n_sensors > (uint32_t)kMaxSensors) return fail(h, RPW_ERR_UNSUPPORTED, "%u sensors > %d", n_sensors, kMaxSensors);
    if (n_sensors && !sensors) return fail(h, RPW_ERR_INVALID, "sensors is null");
    SensorTable t; std::memset(&t, 0, sizeof t);
    t.n = (int)n_sensors;
    for (uint32_t k = 0; k < n_sensors; ++k) {
        // the reference's own expressions (lidar_fusion.cpp:98, 109-111): float radians, cosf / sinf
        const float angle_degrees = sensors[k].rotation_deg;
        t.rotate[k] = std::abs(angle_degrees) > 1e-6f ? 1 : 0;
        float angle_rad = angle_degrees * M_PI / 180.0f;
        t.cosv[k] = std::cos(angle_rad);
        t.sinv[k] = std::sin(angle_rad);
        // d <= ego_radius  <=>  s < min { s : sqrtf(s) > ego_radius }   (sqrtf is correctly rounded and monotone)
        const float r = sensors[k].ego_radius;
        float g;
        if (!(r == r)) g = 0.0f;                              // NaN radius: nothing is <= NaN
        else if (r < 0.0f) g = 0.0f;
        else if (std::isinf(r)) g = r;
        else {
            const float up = std::nextafter(r, std::numeric_limits<float>::infinity());
            float s = (float)((double)up * (double)up);
            if (std::isinf(s)) s = std::numeric_limits<float>::max();
            while (s > 0.0f && std::sqrt(std::nextafter(s, 0.0f)) >= up) s = std::nextafter(s, 0.0f);
            while (!(std::sqrt(s) >= up)) { if (s == std::numeric_limits<float>::max()) { s = std::numeric_limits<float>::infinity(); break; } s = std::nextafter(s, std::numeric_limits<float>::infinity()); }
            g = s;
        }
        t.gt_ego[k] = g;
    }
    h->sensors = t;
    return RPW_OK;
}

rpw_status rpw_segment_fused(rpw_handle* h, const float* xyz, const uint32_t* sensor_offsets,
                             uint8_t* labels, float* residual, rpw_frame_stats* stats)
{
    if (!h) return RPW_ERR_INVALID;
    if (h->sensors.n <= 0) return fail(h, RPW_ERR_INVALID, "rpw_set_sensors has not been called");
    if (!sensor_offsets) return fail(h, RPW_ERR_INVALID, "sensor_offsets is null");
    const uint32_t n = sensor_offsets[h->sensors.n];
    if (n && (!xyz || !labels)) return fail(h, RPW_ERR_INVALID, "null pointer");
    if (n > h->max_points) return fail(h, RPW_ERR_CAPACITY, "frame of %u points > capacity %u", n, h->max_points);
    RPW_HIP(h, hipSetDevice(h->device));
    const size_t N = h->max_points ? h->max_points : 1;
    if (!h->d_in) RPW_HIP(h, hipMalloc(&h->d_in, N * 3 * sizeof(float)));
    if (!h->d_labels) RPW_HIP(h, hipMalloc(&h->d_labels, N));
    if (residual && !h->d_res) RPW_HIP(h, hipMalloc(&h->d_res, N * sizeof(float)));
    if (!h->own_stream) RPW_HIP(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    hipStream_t s = h->own_stream;
    if (n) RPW_HIP(h, hipMemcpyAsync(h->d_in, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, s));
    const uint32_t off[2] = {0, n};
    rpw_status st = run_batch(h, h->ws, h->d_in, off, 1, h->d_labels, residual ? h->d_res : nullptr, s, InputLayout(), sensor_offsets);
    if (st != RPW_OK) return st;
    if (n) RPW_HIP(h, hipMemcpyAsync(labels, h->d_labels, n, hipMemcpyDeviceToHost, s));
    if (n && residual) RPW_HIP(h, hipMemcpyAsync(residual, h->d_res, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s));
    return finish_one_frame(h, s, stats);
}

rpw_status rpw_segment_batch_device_fused(rpw_handle* h, const float* d_xyz, const uint32_t* frame_offsets,
                                          const uint32_t* sensor_offsets, uint32_t n_frames, uint8_t* d_labels,
                                          float* d_residual, void* stream)
{
    if (!h) return RPW_ERR_INVALID;
    if (h->sensors.n <= 0) return fail(h, RPW_ERR_INVALID, "rpw_set_sensors has not been called");
    if (!sensor_offsets) return fail(h, RPW_ERR_INVALID, "sensor_offsets is null");
    return run_batch(h, h->ws, d_xyz, frame_offsets, n_frames, d_labels, d_residual, (hipStream_t)stream, InputLayout(), sensor_offsets);
}

rpw_status rpw_set_trace(rpw_handle* h, int enable, uint32_t max_nodes)
{
    if (!h) return RPW_ERR_INVALID;
    RPW_HIP(h, hipSetDevice(h->device));
    RPW_HIP(h, hipDeviceSynchronize());
    if (h->d_nodes) { (void)hipFree(h->d_nodes); h->d_nodes = nullptr; }
    h->trace = enable ? 1 : 0; h->nodes_cap = enable ? max_nodes : 0;
    if (h->trace && h->nodes_cap) RPW_HIP(h, hipMalloc(&h->d_nodes, (size_t)h->nodes_cap * sizeof(rpw_node)));
    return RPW_OK;
}

rpw_status rpw_get_nodes(rpw_handle* h, rpw_node* nodes, uint32_t cap, uint32_t* n_nodes)
{
    if (!h || !n_nodes) return RPW_ERR_INVALID;
    *n_nodes = 0;
    if (!h->trace || !h->d_nodes) return fail(h, RPW_ERR_INVALID, "trace is not enabled");
    RPW_HIP(h, hipSetDevice(h->device));
    RPW_HIP(h, hipStreamSynchronize(h->ws.last_stream));
    uint32_t cnt = 0;
    RPW_HIP(h, hipMemcpy(&cnt, h->ws.d_ctrl + kBuckets, sizeof(uint32_t), hipMemcpyDeviceToHost));
    *n_nodes = cnt;
    uint32_t m = cnt < h->nodes_cap ? cnt : h->nodes_cap;
    if (m > cap) m = cap;
    if (m && nodes) RPW_HIP(h, hipMemcpy(nodes, h->d_nodes, (size_t)m * sizeof(rpw_node), hipMemcpyDeviceToHost));
    return RPW_OK;
}

rpw_status rpw_set_profiling(rpw_handle* h, int enable)
{
    if (!h) return RPW_ERR_INVALID;
    if (!enable) drain_profile(h);
    h->profiling = enable ? 1 : 0;
    return RPW_OK;
}

rpw_status rpw_get_kernel_ms(rpw_handle* h, double ms[RPW_K_COUNT], uint64_t launches[RPW_K_COUNT], int reset)
{
    if (!h) return RPW_ERR_INVALID;
    drain_profile(h);
    for (int k = 0; k < RPW_K_COUNT; ++k) {
        if (ms) ms[k] = h->ms[k];
        if (launches) launches[k] = h->launches[k];
        if (reset) { h->ms[k] = 0.0; h->launches[k] = 0; }
    }
    return RPW_OK;
}

rpw_status rpw_debug_angles(rpw_handle* h, const float* x, const float* y, float* angle, uint32_t n)
{
    if (!h || (n && (!x || !y || !angle))) return RPW_ERR_INVALID;
    RPW_HIP(h, hipSetDevice(h->device));
    float *dx = nullptr, *dy = nullptr, *da = nullptr;
    const size_t bytes = (size_t)(n ? n : 1) * sizeof(float);
    RPW_HIP(h, hipMalloc(&dx, bytes)); RPW_HIP(h, hipMalloc(&dy, bytes)); RPW_HIP(h, hipMalloc(&da, bytes));
    RPW_HIP(h, hipMemcpy(dx, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    RPW_HIP(h, hipMemcpy(dy, y, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    launch_debug_angles(dx, dy, da, n, nullptr);
    RPW_HIP(h, hipStreamSynchronize(nullptr));
    RPW_HIP(h, hipMemcpy(angle, da, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(da);
    return RPW_OK;
}

rpw_status rpw_debug_eig3(rpw_handle* h, const float* cov9, float* normal3, uint32_t n)
{
    if (!h || (n && (!cov9 || !normal3))) return RPW_ERR_INVALID;
    RPW_HIP(h, hipSetDevice(h->device));
    float *dc = nullptr, *dn = nullptr;
    RPW_HIP(h, hipMalloc(&dc, (size_t)(n ? n : 1) * 9 * sizeof(float)));
    RPW_HIP(h, hipMalloc(&dn, (size_t)(n ? n : 1) * 3 * sizeof(float)));
    RPW_HIP(h, hipMemcpy(dc, cov9, (size_t)n * 9 * sizeof(float), hipMemcpyHostToDevice));
    launch_debug_eig3(dc, dn, n, nullptr);
    RPW_HIP(h, hipStreamSynchronize(nullptr));
    RPW_HIP(h, hipMemcpy(normal3, dn, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(dc); (void)hipFree(dn);
    return RPW_OK;
}

rpw_status rpw_debug_loop_bench(rpw_handle* h, int what, uint32_t n, uint32_t inlier_pct, int reps, uint32_t blocks, double* ms)
{
    return rpw_debug_loop_bench_ticks(h, what, n, inlier_pct, reps, blocks, ms, nullptr);
}

rpw_status rpw_debug_loop_bench_ticks(rpw_handle* h, int what, uint32_t n, uint32_t inlier_pct, int reps, uint32_t blocks, double* ms, double* ticks)
{
    if (!h || !ms || n == 0 || blocks == 0 || reps <= 0 || inlier_pct > 100) return RPW_ERR_INVALID;
    if ((uint64_t)n * blocks > (1ull << 28)) return fail(h, RPW_ERR_CAPACITY, "loop bench: n * blocks too large");
    const int base_what = what & 31;
    if ((what & ~127) || (what & 96) == 32 || !((base_what >= 0 && base_what <= 7) || (base_what >= 16 && base_what <= 18)))
        return fail(h, RPW_ERR_INVALID, "loop bench: unknown loop %d", what);
    RPW_HIP(h, hipSetDevice(h->device));
    const size_t total = (size_t)n * blocks;
    // synthetic node: a plane z ~ 0 with noise and 1 - inlier_pct % of points well above it; the index list holds the inliers
    std::vector<float> x(n), y(n), z(n); std::vector<uint32_t> il(n); std::vector<uint8_t> m(n);
    uint32_t count = 0; uint32_t seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)(seed >> 8) * (1.0f / 16777216.0f); };
    for (uint32_t i = 0; i < n; ++i) {
        x[i] = 20.0f + 10.0f * rnd(); y[i] = -5.0f + 10.0f * rnd();
        const bool in = (uint32_t)(rnd() * 100.0f) < inlier_pct;
        z[i] = in ? 0.02f * (rnd() - 0.5f) : 1.0f + rnd();
        m[i] = in ? 1 : 0;
        if (in) il[count++] = i;
    }
    if (count == 0) { il[0] = 0; count = 1; }
    float *dx = nullptr, *dy = nullptr, *dz = nullptr, *dout = nullptr; unsigned long long* dm = nullptr; uint32_t* dil = nullptr;
    // mask bit planes: node b starts at word (b * n >> 6) + b of every plane; all three planes hold the synthetic mask
    const uint32_t mplane = (uint32_t)(total / 64 + blocks + 8);
    std::vector<unsigned long long> words((size_t)mplane, 0ull);
    for (uint32_t b = 0; b < blocks; ++b) {
        const size_t w0 = (((size_t)b * n) >> 6) + b;
        for (uint32_t i = 0; i < n; ++i) if (m[i]) words[w0 + (i >> 6)] |= 1ull << (i & 63u);
    }
    auto cleanup = [&]() { (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dz); (void)hipFree(dout); (void)hipFree(dm); (void)hipFree(dil); };
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMalloc(&dx, total * 4);
    if (e == hipSuccess) e = hipMalloc(&dy, total * 4);
    if (e == hipSuccess) e = hipMalloc(&dz, total * 4);
    if (e == hipSuccess) e = hipMalloc(&dil, total * 4);
    if (e == hipSuccess) e = hipMalloc(&dm, 3 * (size_t)mplane * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&dout, (size_t)blocks * 4);
    unsigned long long* dticks = nullptr;
    if (e == hipSuccess) e = hipMalloc(&dticks, 8);
    for (uint32_t b = 0; e == hipSuccess && b < blocks; ++b) {
        e = hipMemcpy(dx + (size_t)b * n, x.data(), (size_t)n * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dy + (size_t)b * n, y.data(), (size_t)n * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dz + (size_t)b * n, z.data(), (size_t)n * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dil + (size_t)b * n, il.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    }
    for (int k = 0; e == hipSuccess && k < 3; ++k) e = hipMemcpy(dm + (size_t)k * mplane, words.data(), (size_t)mplane * sizeof(unsigned long long), hipMemcpyHostToDevice);
    hipEvent_t ea = nullptr, eb = nullptr;
    if (e == hipSuccess) e = hipEventCreate(&ea);
    if (e == hipSuccess) e = hipEventCreate(&eb);
    float t = 0.0f;
    if (e == hipSuccess) {
        const uint32_t cnt = base_what < 16 ? (base_what == kSweepRangeMode || base_what >= 4 ? n : count) : n;
        launch_loop_bench(dx, dy, dz, dm, mplane, dil, n, cnt, what, 1, blocks, dout, dticks, nullptr);            // warm-up
        e = hipEventRecord(ea, nullptr);
        launch_loop_bench(dx, dy, dz, dm, mplane, dil, n, cnt, what, reps, blocks, dout, dticks, nullptr);
        if (e == hipSuccess) e = hipEventRecord(eb, nullptr);
        if (e == hipSuccess) e = hipEventSynchronize(eb);
        if (e == hipSuccess) e = hipEventElapsedTime(&t, ea, eb);
    }
    unsigned long long hticks = 0;
    if (e == hipSuccess) e = hipMemcpy(&hticks, dticks, 8, hipMemcpyDeviceToHost);
    if (ticks) *ticks = (double)hticks;
    (void)hipFree(dticks);
    if (ea) (void)hipEventDestroy(ea);
    if (eb) (void)hipEventDestroy(eb);
    cleanup();
    if (e != hipSuccess) return fail(h, RPW_ERR_HIP, "loop bench: %s", hipGetErrorString(e));
    *ms = (double)t;
    return RPW_OK;
}

void rpw_debug_angles_host(const float* x, const float* y, float* angle, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i) angle[i] = rpw::polar_angle(x[i], y[i]);
}

} // extern "C"
