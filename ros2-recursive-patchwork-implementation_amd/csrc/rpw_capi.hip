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

constexpr int kStageSlots = 8;

struct EventPair { int kernel; hipEvent_t a, b; };

} // namespace

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

    // batch buffers
    uint32_t* d_offsets = nullptr;       // [2*(F+1)] frame_off then tile_base, then [F][kMaxSensors+1] sensor offsets
    uint32_t* h_offsets[kStageSlots] = {};   // pinned staging ring
    hipEvent_t stage_ev[kStageSlots] = {};
    bool stage_used[kStageSlots] = {};
    int stage_next = 0;

    uint32_t* d_hist = nullptr;
    uint32_t max_tiles = 0;
    uint32_t* d_patch_begin = nullptr;
    PatchDesc* d_patches = nullptr;
    uint32_t* d_ctrl = nullptr;          // [kBuckets] bucket counts + [1] node count (+pad), then [F][PB] class totals of every frame
    uint32_t* d_bucket_list = nullptr;
    float *d_sx = nullptr, *d_sy = nullptr, *d_sz = nullptr, *d_ax = nullptr, *d_ay = nullptr, *d_az = nullptr;
    float4* d_tp = nullptr;
    uint32_t* d_tdir = nullptr;
    uint32_t* d_sidx = nullptr;
    uint8_t* d_mask = nullptr;
    rpw_frame_stats* d_stats = nullptr;

    // trace
    int trace = 0;
    uint32_t nodes_cap = 0;
    rpw_node* d_nodes = nullptr;

    // host-pointer path
    hipStream_t own_stream = nullptr;
    float* d_in = nullptr;
    unsigned char* d_raw = nullptr;      // raw PointCloud2 bytes (rpw_segment_pointcloud2)
    size_t raw_bytes = 0;
    uint8_t* d_labels = nullptr;
    float* d_res = nullptr;

    // last batch
    hipStream_t last_stream = nullptr;
    uint32_t last_frames = 0;

    // profiling
    int profiling = 0;
    std::vector<EventPair> pending;
    std::vector<hipEvent_t> free_events;
    double ms[RPW_K_COUNT] = {0, 0, 0};
    uint64_t launches[RPW_K_COUNT] = {0, 0, 0};
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

void free_device(rpw_handle* h)
{
    auto F = [](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
    F(h->d_bounds); F(h->d_offsets); F(h->d_hist); F(h->d_patch_begin); F(h->d_patches);
    F(h->d_ctrl); F(h->d_bucket_list); F(h->d_sx); F(h->d_sy); F(h->d_sz); F(h->d_ax); F(h->d_ay); F(h->d_az);
    F(h->d_sidx); F(h->d_mask); F(h->d_tp); F(h->d_tdir); F(h->d_stats); F(h->d_nodes); F(h->d_in); F(h->d_raw); F(h->d_labels); F(h->d_res);
    h->raw_bytes = 0;
    for (int i = 0; i < kStageSlots; ++i) {
        if (h->h_offsets[i]) { (void)hipHostFree(h->h_offsets[i]); h->h_offsets[i] = nullptr; }
        if (h->stage_ev[i]) { (void)hipEventDestroy(h->stage_ev[i]); h->stage_ev[i] = nullptr; }
        h->stage_used[i] = false;
    }
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

rpw_status allocate(rpw_handle* h)
{
    free_device(h);
    const size_t N = h->max_points ? h->max_points : 1, F = h->max_frames ? h->max_frames : 1;
    h->S = h->cfg.num_sectors; h->P = 8 * h->S; h->PB = (h->P + 4 + 3) & ~3;       // patch classes + {no patch, outside, non-finite, ego-removed}
    std::vector<float> bounds;
    build_tables(h->cfg, h->tab, bounds);
    RPW_HIP(h, hipMalloc(&h->d_bounds, bounds.size() * sizeof(float)));
    RPW_HIP(h, hipMemcpy(h->d_bounds, bounds.data(), bounds.size() * sizeof(float), hipMemcpyHostToDevice));
    const size_t off_words = 2 * (F + 1) + F * (kMaxSensors + 1);
    RPW_HIP(h, hipMalloc(&h->d_offsets, off_words * sizeof(uint32_t)));
    for (int i = 0; i < kStageSlots; ++i) {
        RPW_HIP(h, hipHostMalloc((void**)&h->h_offsets[i], off_words * sizeof(uint32_t), hipHostMallocDefault));
        RPW_HIP(h, hipEventCreateWithFlags(&h->stage_ev[i], hipEventDisableTiming));
    }
    h->max_tiles = (uint32_t)(N / kTilePoints + F);
    RPW_HIP(h, hipMalloc(&h->d_hist, (size_t)h->max_tiles * h->PB * sizeof(uint32_t)));
    RPW_HIP(h, hipMalloc(&h->d_patch_begin, F * (size_t)(h->P + 1) * sizeof(uint32_t)));
    RPW_HIP(h, hipMalloc(&h->d_patches, F * (size_t)h->P * sizeof(PatchDesc)));
    RPW_HIP(h, hipMalloc(&h->d_ctrl, (kBuckets + 4 + F * (size_t)h->PB) * sizeof(uint32_t)));     // + per-frame class totals
    RPW_HIP(h, hipMalloc(&h->d_bucket_list, (size_t)kBuckets * F * h->P * sizeof(uint32_t)));
    RPW_HIP(h, hipMalloc(&h->d_sx, N * sizeof(float)));
    RPW_HIP(h, hipMalloc(&h->d_sy, N * sizeof(float)));
    RPW_HIP(h, hipMalloc(&h->d_sz, N * sizeof(float)));
    RPW_HIP(h, hipMalloc(&h->d_ax, N * sizeof(float)));
    RPW_HIP(h, hipMalloc(&h->d_ay, N * sizeof(float)));
    RPW_HIP(h, hipMalloc(&h->d_az, N * sizeof(float)));
    RPW_HIP(h, hipMalloc(&h->d_sidx, N * sizeof(uint32_t)));
    {   // tile-sorted records: slot < frame offset + frame size, so N records suffice
        RPW_HIP(h, hipMalloc(&h->d_tp, N * sizeof(float4)));
        RPW_HIP(h, hipMalloc(&h->d_tdir, (size_t)h->max_tiles * h->PB * sizeof(uint32_t)));
    }
    RPW_HIP(h, hipMalloc(&h->d_mask, N));
    RPW_HIP(h, hipMalloc(&h->d_stats, F * sizeof(rpw_frame_stats)));
    if (h->trace && h->nodes_cap) RPW_HIP(h, hipMalloc(&h->d_nodes, (size_t)h->nodes_cap * sizeof(rpw_node)));
    return RPW_OK;
}

hipEvent_t get_event(rpw_handle* h)
{
    if (!h->free_events.empty()) { hipEvent_t e = h->free_events.back(); h->free_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
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
    (void)hipEventRecord(ep.a, s);
    launch();
    (void)hipEventRecord(ep.b, s);
    h->pending.push_back(ep);
    if (h->pending.size() > 8192) drain_profile(h);
}

struct InputLayout { uint32_t stride = 12, off[3] = {0, 4, 8}; };

rpw_status run_batch(rpw_handle* h, const void* d_xyz, const uint32_t* frame_offsets, uint32_t n_frames,
                     uint8_t* d_labels, float* d_residual, hipStream_t stream, const InputLayout& lay = InputLayout(),
                     const uint32_t* sensor_offsets = nullptr /* [F][sensors.n + 1], frame-relative */)
{
    if (!frame_offsets) return fail(h, RPW_ERR_INVALID, "frame_offsets is null");
    if (n_frames > h->max_frames) return fail(h, RPW_ERR_CAPACITY, "n_frames %u > capacity %u", n_frames, h->max_frames);
    h->last_stream = stream; h->last_frames = n_frames;
    if (n_frames == 0) return RPW_OK;
    const uint32_t total = frame_offsets[n_frames];
    if (total > h->max_points) return fail(h, RPW_ERR_CAPACITY, "batch of %u points > capacity %u", total, h->max_points);
    if (total && (!d_xyz || !d_labels)) return fail(h, RPW_ERR_INVALID, "null device pointer");
    RPW_HIP(h, hipSetDevice(h->device));

    const int slot = h->stage_next; h->stage_next = (h->stage_next + 1) % kStageSlots;
    if (h->stage_used[slot]) RPW_HIP(h, hipEventSynchronize(h->stage_ev[slot]));
    uint32_t* ho = h->h_offsets[slot];
    const uint32_t F = n_frames;
    uint32_t tiles = 0;
    for (uint32_t f = 0; f <= F; ++f) {
        if (f && frame_offsets[f] < frame_offsets[f - 1]) return fail(h, RPW_ERR_INVALID, "frame_offsets not monotone");
        ho[f] = frame_offsets[f];
        ho[(F + 1) + f] = tiles;
        if (f < F) tiles += (frame_offsets[f + 1] - frame_offsets[f] + kTilePoints - 1) / kTilePoints;
    }
    if (ho[0] != 0) return fail(h, RPW_ERR_INVALID, "frame_offsets[0] must be 0");
    const int ns = sensor_offsets ? h->sensors.n : 0;
    size_t copy_words = 2 * (size_t)(F + 1);
    if (ns > 0) {
        uint32_t* so = ho + 2 * (F + 1);
        for (uint32_t f = 0; f < F; ++f) {
            const uint32_t nf = frame_offsets[f + 1] - frame_offsets[f];
            const uint32_t* src = sensor_offsets + (size_t)f * (ns + 1);
            if (src[0] != 0 || src[ns] != nf) return fail(h, RPW_ERR_INVALID, "sensor offsets of frame %u must run from 0 to the frame size", f);
            for (int j = 0; j <= ns; ++j) {
                if (j && src[j] < src[j - 1]) return fail(h, RPW_ERR_INVALID, "sensor offsets of frame %u not monotone", f);
                so[(size_t)f * (ns + 1) + j] = src[j];
            }
        }
        copy_words += (size_t)F * (ns + 1);
    }
    if (tiles > h->max_tiles) return fail(h, RPW_ERR_CAPACITY, "tile count %u > capacity %u", tiles, h->max_tiles);
    for (uint32_t f = 0; f < F; ++f)      // bin_tile addresses a frame's records with 32-bit byte offsets
        if ((uint64_t)(frame_offsets[f + 1] - frame_offsets[f]) * 16u > 0xffffffffull)
            return fail(h, RPW_ERR_CAPACITY, "frame %u has %u points; the limit is 268435455 per frame", f, frame_offsets[f + 1] - frame_offsets[f]);
    RPW_HIP(h, hipMemcpyAsync(h->d_offsets, ho, copy_words * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    RPW_HIP(h, hipEventRecord(h->stage_ev[slot], stream));
    h->stage_used[slot] = true;
    RPW_HIP(h, hipMemsetAsync(h->d_ctrl, 0, (kBuckets + 4 + (size_t)F * h->PB) * sizeof(uint32_t), stream));

    DevParams p;
    p.cfg = h->cfg; p.tab = h->tab; p.bounds = h->d_bounds; p.S = h->S; p.P = h->P; p.PB = h->PB;
    p.n_frames = F; p.n_tiles = tiles; p.frame_first = 0; p.frame_count = F; p.tile_first = 0; p.tile_count = tiles; p.frame_off = h->d_offsets; p.tile_base = h->d_offsets + (F + 1);
    p.xyz = reinterpret_cast<const P3*>(d_xyz); p.in_stride = lay.stride;
    p.in_off[0] = lay.off[0]; p.in_off[1] = lay.off[1]; p.in_off[2] = lay.off[2]; p.labels = d_labels;
    p.sensors = h->sensors; if (ns == 0) p.sensors.n = 0;
    p.sensor_off = h->d_offsets + 2 * (F + 1); p.residual = d_residual;
    p.hist = h->d_hist; p.ftot = h->d_ctrl + kBuckets + 4; p.patch_begin = h->d_patch_begin; p.patches = h->d_patches;
    p.bucket_count = h->d_ctrl; p.bucket_list = h->d_bucket_list; p.list_stride = h->max_frames * (uint32_t)h->P;
    p.tp = h->d_tp; p.tdir = h->d_tdir;
    p.sx = h->d_sx; p.sy = h->d_sy; p.sz = h->d_sz; p.sidx = h->d_sidx; p.ax = h->d_ax; p.ay = h->d_ay; p.az = h->d_az;
    p.mask = h->d_mask; p.stats = h->d_stats;
    p.trace = (h->trace && h->d_nodes) ? 1 : 0; p.nodes = h->d_nodes; p.node_count = h->d_ctrl + kBuckets; p.nodes_cap = h->nodes_cap;

    // ---- binning pass: K1 and K2 once over the whole batch, then K3 ---------------------------------------------
    // (A variant that cut large batches into groups of frames and pipelined K1 / K2 over two streams, so that K3
    // would find the xyz still in the Infinity Cache, was measured slower -- launch-bound -- and removed.)
    timed(h, RPW_K_BIN, stream, [&] { launch_bin(p, stream); });
    timed(h, RPW_K_SCAN, stream, [&] { launch_scan(p, stream); });
    timed(h, RPW_K_FIT, stream, [&] { launch_fit(p, stream); });
    RPW_HIP(h, hipGetLastError());
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
    if (max_points > (1u << 30))      // patch_fit addresses its per-slot arrays with 32-bit byte offsets
        return fail(nullptr, RPW_ERR_CAPACITY, "max_points %u > 1073741824", max_points);
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
    return run_batch(h, d_xyz, frame_offsets, n_frames, d_labels, d_residual, (hipStream_t)stream);
}

rpw_status rpw_get_batch_stats(rpw_handle* h, rpw_frame_stats* stats, uint32_t n_frames)
{
    if (!h || !stats) return RPW_ERR_INVALID;
    if (n_frames > h->last_frames) return fail(h, RPW_ERR_INVALID, "asked for %u frames, last batch had %u", n_frames, h->last_frames);
    RPW_HIP(h, hipSetDevice(h->device));
    RPW_HIP(h, hipStreamSynchronize(h->last_stream));
    if (n_frames) RPW_HIP(h, hipMemcpy(stats, h->d_stats, n_frames * sizeof(rpw_frame_stats), hipMemcpyDeviceToHost));
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
    rpw_status st = run_batch(h, h->d_in, off, 1, h->d_labels, residual ? h->d_res : nullptr, s);
    if (st != RPW_OK) return st;
    if (n) RPW_HIP(h, hipMemcpyAsync(labels, h->d_labels, n, hipMemcpyDeviceToHost, s));
    if (n && residual) RPW_HIP(h, hipMemcpyAsync(residual, h->d_res, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s));
    RPW_HIP(h, hipStreamSynchronize(s));
    if (stats) RPW_HIP(h, hipMemcpy(stats, h->d_stats, sizeof(rpw_frame_stats), hipMemcpyDeviceToHost));
    return RPW_OK;
}

rpw_status rpw_segment_pointcloud2(rpw_handle* h, const uint8_t* data, uint32_t n, uint32_t point_step,
                                   uint32_t off_x, uint32_t off_y, uint32_t off_z,
                                   uint8_t* labels, float* residual, rpw_frame_stats* stats)
{
    if (!h) return RPW_ERR_INVALID;
    if (n && (!data || !labels)) return fail(h, RPW_ERR_INVALID, "null pointer");
    if (point_step < 12 || (point_step & 3u) || (off_x & 3u) || (off_y & 3u) || (off_z & 3u) ||
        off_x + 4 > point_step || off_y + 4 > point_step || off_z + 4 > point_step)
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
    rpw_status st = run_batch(h, h->d_raw, off, 1, h->d_labels, residual ? h->d_res : nullptr, s, lay);
    if (st != RPW_OK) return st;
    if (n) RPW_HIP(h, hipMemcpyAsync(labels, h->d_labels, n, hipMemcpyDeviceToHost, s));
    if (n && residual) RPW_HIP(h, hipMemcpyAsync(residual, h->d_res, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s));
    RPW_HIP(h, hipStreamSynchronize(s));
    if (stats) RPW_HIP(h, hipMemcpy(stats, h->d_stats, sizeof(rpw_frame_stats), hipMemcpyDeviceToHost));
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
    rpw_status st = run_batch(h, h->d_in, off, 1, h->d_labels, residual ? h->d_res : nullptr, s, InputLayout(), sensor_offsets);
    if (st != RPW_OK) return st;
    if (n) RPW_HIP(h, hipMemcpyAsync(labels, h->d_labels, n, hipMemcpyDeviceToHost, s));
    if (n && residual) RPW_HIP(h, hipMemcpyAsync(residual, h->d_res, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s));
    RPW_HIP(h, hipStreamSynchronize(s));
    if (stats) RPW_HIP(h, hipMemcpy(stats, h->d_stats, sizeof(rpw_frame_stats), hipMemcpyDeviceToHost));
    return RPW_OK;
}

rpw_status rpw_segment_batch_device_fused(rpw_handle* h, const float* d_xyz, const uint32_t* frame_offsets,
                                          const uint32_t* sensor_offsets, uint32_t n_frames, uint8_t* d_labels,
                                          float* d_residual, void* stream)
{
    if (!h) return RPW_ERR_INVALID;
    if (h->sensors.n <= 0) return fail(h, RPW_ERR_INVALID, "rpw_set_sensors has not been called");
    if (!sensor_offsets) return fail(h, RPW_ERR_INVALID, "sensor_offsets is null");
    return run_batch(h, d_xyz, frame_offsets, n_frames, d_labels, d_residual, (hipStream_t)stream, InputLayout(), sensor_offsets);
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
    RPW_HIP(h, hipStreamSynchronize(h->last_stream));
    uint32_t cnt = 0;
    RPW_HIP(h, hipMemcpy(&cnt, h->d_ctrl + kBuckets, sizeof(uint32_t), hipMemcpyDeviceToHost));
    *n_nodes = cnt;
#if defined(RPW_STAMPS)
    cnt = h->nodes_cap;   /* diagnostic build: return the whole buffer (stamps sit in its tail) */
#endif
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

void rpw_debug_angles_host(const float* x, const float* y, float* angle, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i) angle[i] = rpw::polar_angle(x[i], y[i]);
}

} // extern "C"
