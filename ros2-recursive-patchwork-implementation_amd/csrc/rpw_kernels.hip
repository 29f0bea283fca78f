// rpw_kernels.hip -- hand-written gfx950 kernels of the Recursive Patchwork hot path.
//
//   K1 bin_tile        cleanPoints + computeDistances2D + filterPointsByRadius + computeAngles +
//                      patch membership (recursive_patchwork.cpp:315-378), O(N) instead of O(P*N), as a stable counting
//                      sort by patch in two launches around K2: PASS 0 counts every tile's classes (+ the frame's class
//                      totals, integer atomics), PASS 1 ranks the tile's points and writes them straight to their final,
//                      patch-sorted slots (recursive_patchwork.cpp:372-378: patch_points in zone order)
//   K2 scan_offsets    per frame: one scan over the classes -> patch begins, patch descriptors, size-bucket lists, and the
//                      first destination slot of every (tile, class)
//   K3 patch_fit       fitPlaneAndSplit / fitPlanePCA / computeCentroid / computeCovariance /
//                      Eigen solve / computePlaneDistances / mask compare / split + recursion
//                      (recursive_patchwork.cpp:77-308). One OWNER wave per (frame, patch) runs the passes; the dependent
//                      adds of its sequential sums run on the owner itself (patch_fit1_kernel), on a server wave that adds
//                      for seven owners at once (patch_fit_srv_kernel, batches), or on a server wave of the patch's own
//                      (patch_fit_srv1_kernel, launches of few patches) -- see launch_fit and DESIGN.md section 4
//                      The sweeps over a mask's inliers take them from an index list the fused pass wrote (the first and
//                      the last form) or expand the mask's bit plane themselves (the batch form: IdxGen).
//
// Exactness: every sum the reference accumulates sequentially in binary32 (centroid,
// covariance, residual, mean distance, variance) is accumulated here in the same order
// by one lane per accumulator; the other lanes only prepare the addends.
// Built with -ffp-contract=off; divide and sqrt are correctly rounded.
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdlib>
#include <mutex>
#include <set>
#include <utility>
#include "rpw_internal.h"

namespace rpw {

// ------------------------------------------------------------------------------------------
// wavefront helpers (64 lanes)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __ballot(p); }
__device__ __forceinline__ int mbcnt64(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}
__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(v, o); if (l >= o) v += t; }
    return v;
}

__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* red /*[5]*/, uint32_t& total)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t incl = wave_incl_scan(v);
    if (lane == 63) red[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const uint32_t t = red[w]; if (w < wave) base += t; }
    total = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return base + incl - v;
}


// ------------------------------------------------------------------------------------------
// input record: packed Point3D (12 B, one dwordx3 load) or a PointCloud2-style record (x, y, z FLOAT32 at
// byte offsets inside a point_step-byte record: three dword loads)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ P3 load_point(const DevParams& p, uint32_t i)
{
    if (p.in_stride == 12u) return p.xyz[i];
    const unsigned char* rec = reinterpret_cast<const unsigned char*>(p.xyz) + (size_t)i * p.in_stride;
    P3 q;
    q.x = *reinterpret_cast<const float*>(rec + p.in_off[0]);
    q.y = *reinterpret_cast<const float*>(rec + p.in_off[1]);
    q.z = *reinterpret_cast<const float*>(rec + p.in_off[2]);
    return q;
}

// LidarFusion pre-step for one point of frame f (index i inside the frame): yaw rotation of its sensor, then
// the ego-radius test. Returns true when the point is removed (inside the ego radius).
__device__ __forceinline__ bool fuse_point(const DevParams& p, uint32_t f, uint32_t i, P3& q)
{
    const SensorTable& t = p.sensors;
    const uint32_t* so = p.sensor_off + (size_t)f * (t.n + 1);
    int k = 0;
    for (int j = 1; j < t.n; ++j) k += (i >= so[j]) ? 1 : 0;
    float c = t.cosv[0], s = t.sinv[0], g = t.gt_ego[0]; int rot = t.rotate[0];
#pragma unroll
    for (int j = 1; j < kMaxSensors; ++j) if (j < t.n && k == j) { c = t.cosv[j]; s = t.sinv[j]; g = t.gt_ego[j]; rot = t.rotate[j]; }
    if (rot) {
        const float rx = q.x * c - q.y * s;
        const float ry = q.x * s + q.y * c;
        q.x = rx; q.y = ry;
    }
    return (q.x * q.x + q.y * q.y) < g;
}

// ------------------------------------------------------------------------------------------
// tile -> frame lookup
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t frame_of_tile(const uint32_t* __restrict__ tile_base, uint32_t n_frames, uint32_t tile)
{
    uint32_t lo = 0, hi = n_frames;          // tile_base[lo] <= tile < tile_base[hi]
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (tile_base[mid] <= tile) lo = mid; else hi = mid; }
    return lo;
}

// ------------------------------------------------------------------------------------------
// K1: the binning pass, two launches of one kernel around K2 (a counting sort by patch, stable):
//   PASS 0 (bin_count)    every tile of 2048 points is read (12 B/pt) and classified; the tile's count of every class goes
//                         to hist[tile][class] and, with integer atomics, into the frame's class totals (ftot)
//   K2 (scan_offsets)     turns the counts into destinations: hist[tile][class] = first slot, relative to the frame, of
//                         this tile's points of that class in the patch-sorted arrays
//   PASS 1 (bin_scatter)  the tile is read and classified again (the same deterministic arithmetic), its points are ranked
//                         inside the tile (stable: wave-contiguous sub-tiles, ballot/mbcnt ranks, exclusive prefix over the
//                         4 waves) and written STRAIGHT to their final slots sx / sy / sz / sidx (16 B/pt): patch by patch,
//                         input order kept inside a patch (recursive_patchwork.cpp:372-378: patch_points in zone order)
// 40 bytes per point through HBM for "radial-sector scatter + per-bin prefix-sum compaction". (Rounds 1-2 wrote tile-sorted
// 16-byte records in ONE pass and let patch_fit gather each patch's runs: 28 + 32 = 60 bytes per point, the gather's 32 at
// patch_fit's low rate. A one-pass scatter would need every tile of a frame resident at once to wait for the frame's class
// totals -- a spin on other workgroups that concurrent launches on other streams can starve; not built.)
// ------------------------------------------------------------------------------------------
template <bool STRIDED, bool FUSED, int PASS>
__device__ __forceinline__ void bin_tile_body(const DevParams& p, uint32_t* lds_u32)
{
    uint32_t* wcnt = lds_u32;                                     // [4][PB] per-wave running counts, then (PASS 1) the wave's first slot
    float* bnd = reinterpret_cast<float*>(lds_u32 + 4 * p.PB);    // [S+1] sector bounds
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int j = tid; j < 4 * p.PB; j += kTileThreads) wcnt[j] = 0;
    for (int j = tid; j <= p.S; j += kTileThreads) bnd[j] = p.bounds[j];
    __syncthreads();

    // Workgroups b and b + 8 share an XCD (observed round-robin placement; speed only, never correctness): give every XCD
    // a contiguous range of tiles, so that the adjacent runs consecutive tiles write meet in ONE L2 and leave it as whole lines.
    uint32_t bid = blockIdx.x;
    if (p.bin_remap) {
        const uint32_t q = p.tile_count / 8u, r = p.tile_count % 8u, x = bid % 8u;
        bid = (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + bid / 8u;
    }
    const uint32_t tile = bid + p.tile_first;
    const uint32_t f = frame_of_tile(p.tile_base, p.n_frames, tile);
    const uint32_t fo = p.frame_off[f];
    const uint32_t nf = p.frame_off[f + 1] - fo;
    const uint32_t t_local = tile - p.tile_base[f];
    const uint32_t wbase = t_local * kTilePoints + wave * kWavePoints;
    uint32_t* mycnt = wcnt + wave * p.PB;

    constexpr int kRankSteps = 2;                                  // distinct classes of a row ranked one by one before the match-any path
    const int class_bits = 32 - __clz(p.P + 3);                    // bits that tell the classes 0 .. P+3 apart
    // all of the wave's loads first (8 x 12 B per lane in flight), then the arithmetic
    const char* frame_bytes = reinterpret_cast<const char*>(p.xyz + fo);      // frames are < 4 GB (rpw_create checks)
    P3 q[kTileItems];
#pragma unroll
    for (int r = 0; r < kTileItems; ++r) {
        const uint32_t i = wbase + r * kWave + lane;
        q[r].x = q[r].y = q[r].z = 1.0f;
        // packed: one dwordx3 load, uniform frame base + 32-bit byte offset (no 64-bit address arithmetic per lane)
        if (i < nf) q[r] = STRIDED ? load_point(p, fo + i) : *reinterpret_cast<const P3*>(frame_bytes + i * 12u);
    }
    int cls[kTileItems];
    uint32_t loff[kTileItems];
#pragma unroll
    for (int r = 0; r < kTileItems; ++r) {
        const uint32_t i = wbase + r * kWave + lane;
        const bool valid = i < nf;
        bool ego = false;
        if (FUSED) ego = valid && fuse_point(p, f, i, q[r]);   // q[r] now holds the rotated point
        bool unsure = false;
        int c = classify_point_fast(q[r].x, q[r].y, q[r].z, p.tab, bnd, unsure);
        unsure = unsure && valid && !ego;
        if (ballot64(unsure) != 0ull) {                    // rare: some lane sits within 1e-4 rad of a sector bound
            if (unsure) c = classify_point_exact(q[r].x, q[r].y, q[r].z, p.tab, bnd);
        }
        if (FUSED && ego) c = p.P + 3;                     // removed by the fusion step: in no cloud at all
        cls[r] = valid ? c : p.P;
        loff[r] = 0;
        if (PASS == 1 && valid) {
            // points of a patch get their label (0 or 1) from patch_fit; only the special classes are labelled here
            if (c >= p.P) p.labels[fo + i] = (uint8_t)(c == p.P ? RPW_LABEL_NONGROUND : (c == p.P + 1 ? RPW_LABEL_OUTSIDE : (c == p.P + 2 ? RPW_LABEL_DROPPED : RPW_LABEL_EGO)));
            if (p.residual) p.residual[fo + i] = -1.0f;
        }
        // count (PASS 0) / stable rank (PASS 1) inside the wave's sub-tile: one step per distinct class of the row
        const unsigned long long vmask = ballot64(valid);
        unsigned long long rem = vmask;
        for (int steps = 0; rem && steps < kRankSteps; ++steps) {
            const int first = __ffsll((long long)rem) - 1;
            const int c0 = __builtin_amdgcn_readlane(cls[r], first);      // `first` is wave-uniform: no LDS round trip
            const bool same = cls[r] == c0;                                // (an out-of-range lane may match: its offset is never used)
            const unsigned long long m = ballot64(same) & vmask;
            if (PASS == 1) {
                const uint32_t old = mycnt[c0];
                loff[r] = same ? old + (uint32_t)mbcnt64(m) : loff[r];
                mycnt[c0] = old + (uint32_t)__popcll(m);                   // every lane stores the same value: no exec juggling
            } else if (lane == 0) {
                atomicAdd(&mycnt[c0], (uint32_t)__popcll(m));              // counting only: an LDS add that returns nothing (no round trip)
            }
            rem &= ~m;
        }
        if (rem) {
            // A row of a cloud in RANDOM point order holds dozens of classes: the lanes that are left find their class
            // mates all at once, bit by bit of the class id (a match-any built from ballots), at a cost that does not
            // depend on the number of classes. The first lane of every class bumps the wave's counter (distinct classes,
            // distinct addresses) and hands the run start to its mates.
            const bool todo = (rem >> lane) & 1ull;
            unsigned long long mm = rem;
            for (int b = 0; b < class_bits; ++b) {
                const bool bit = (cls[r] >> b) & 1;
                const unsigned long long bb = ballot64(bit);
                mm &= bit ? bb : ~bb;
            }
            const uint32_t rank = (uint32_t)mbcnt64(mm);
            uint32_t run = 0;
            if (PASS == 0) { if (todo && rank == 0) atomicAdd(&mycnt[cls[r]], (uint32_t)__popcll(mm)); }
            else if (todo && rank == 0) { run = mycnt[cls[r]]; mycnt[cls[r]] = run + (uint32_t)__popcll(mm); }
            if (PASS == 1) {
                run = (uint32_t)__shfl((int)run, __ffsll((long long)mm) - 1);
                loff[r] = todo ? run + rank : loff[r];
            }
        }
    }
    __syncthreads();
    uint32_t* hist_row = p.hist + (size_t)tile * p.PB;
    if (PASS == 0) {
        // the tile's count of every class (patch classes and the four special ones), and the frame's totals
        uint32_t* ftot = p.ftot + (size_t)f * p.PB;                 // integer adds: order-free
        for (int c = tid; c < p.P + 4; c += kTileThreads) {
            const uint32_t v = wcnt[c] + wcnt[p.PB + c] + wcnt[2 * p.PB + c] + wcnt[3 * p.PB + c];
            hist_row[c] = v;
            if (v) atomicAdd(&ftot[c], v);                          // a tile of scan-ordered points touches a few classes only
        }
        return;
    }
    // PASS 1: first slot of every (wave, class) of this tile = the tile's destination of that class (K2) + the waves before
    for (int c = tid; c < p.P; c += kTileThreads) {
        uint32_t s = hist_row[c];
#pragma unroll
        for (int w = 0; w < 4; ++w) { const uint32_t t = wcnt[w * p.PB + c]; wcnt[w * p.PB + c] = s; s += t; }
    }
    __syncthreads();
    char* fx = reinterpret_cast<char*>(p.sx + fo);                  // uniform frame bases + 32-bit byte offsets
    char* fy = reinterpret_cast<char*>(p.sy + fo);
    char* fz = reinterpret_cast<char*>(p.sz + fo);
    char* fi = reinterpret_cast<char*>(p.sidx + fo);
#pragma unroll
    for (int r = 0; r < kTileItems; ++r) {
        const uint32_t i = wbase + r * kWave + lane;
        if (i < nf && cls[r] < p.P) {
            const uint32_t b = (mycnt[cls[r]] + loff[r]) * 4u;       // slot inside the frame's region, in bytes
            *reinterpret_cast<float*>(fx + b) = q[r].x;
            *reinterpret_cast<float*>(fy + b) = q[r].y;
            *reinterpret_cast<float*>(fz + b) = q[r].z;
            *reinterpret_cast<uint32_t*>(fi + b) = i;
        }
    }
}

// (two kernels with names of their own, so that profiles tell the passes apart)
template <bool STRIDED, bool FUSED>
__global__ __launch_bounds__(kTileThreads) void bin_count_kernel(DevParams p)
{
    extern __shared__ uint32_t lds_u32[];
    bin_tile_body<STRIDED, FUSED, 0>(p, lds_u32);
}
template <bool STRIDED, bool FUSED>
__global__ __launch_bounds__(kTileThreads) void bin_scatter_kernel(DevParams p)
{
    extern __shared__ uint32_t lds_u32[];
    bin_tile_body<STRIDED, FUSED, 1>(p, lds_u32);
}

// ------------------------------------------------------------------------------------------
// K2: per-frame offsets. One 256-thread block per frame.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int size_bucket(uint32_t n)
{
    // two buckets per power of two: 2*floor(log2 n) + next bit  (n >= 1)
    const int lg = 31 - __clz((int)n);
    const int half = lg > 0 ? (int)((n >> (lg - 1)) & 1u) : 0;
    int b = 2 * lg + half;
    return b < kBuckets ? b : kBuckets - 1;
}

constexpr int kScanThreads = 256;     // small blocks: a 1024-thread block cannot be placed while patch_fit waves of other streams fill the CUs

__device__ __forceinline__ uint32_t scan_block_excl_scan(uint32_t v, uint32_t* red /*[kScanThreads / 64]*/, uint32_t& total)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t incl = wave_incl_scan(v);
    if (lane == 63) red[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) { const uint32_t t = red[w]; if (w < wave) base += t; tot += t; }
    total = tot;
    __syncthreads();
    return base + incl - v;
}

__global__ __launch_bounds__(kScanThreads) void scan_offsets_kernel(DevParams p)
{
    // One block per frame. The class totals of the frame are there already (bin_tile's counting pass added them up): one scan
    // over the classes gives the patch begins, the patch descriptors and the size-bucket lists. Then the per-tile counts
    // hist[tile][class] become the destinations of bin_tile's scatter pass -- patch begin + the points of that class in the
    // tiles before, in tile (= input) order: class c is walked by `parts` threads, each over a contiguous range of tiles.
    extern __shared__ uint32_t lds_u32[];
    const int C = p.P + 4;
    int parts = kScanThreads / C; if (parts > 16) parts = 16; if (parts < 1) parts = 1;
    uint32_t* tot = lds_u32;                  // [PB] per-class totals, then patch_begin
    uint32_t* lrank = lds_u32 + p.PB;         // [PB] per patch: size bucket << 16 | rank inside the block's share of it
    uint32_t* part = lds_u32 + 2 * p.PB;      // [parts][PB] partial totals, then exclusive prefix over parts
    __shared__ uint32_t red[kScanThreads / 64];
    __shared__ uint32_t bcount[kBuckets];
    const int tid = threadIdx.x;
    const uint32_t f = blockIdx.x + p.frame_first;
    const uint32_t t0 = p.tile_base[f];
    const uint32_t nt = p.tile_base[f + 1] - t0;
    const uint32_t fo = p.frame_off[f];
    const uint32_t nf = p.frame_off[f + 1] - fo;
    uint32_t* base = p.hist + (size_t)t0 * p.PB;
    const uint32_t per = (nt + parts - 1) / parts;

    for (int c = tid; c < C; c += kScanThreads) tot[c] = p.ftot[(size_t)f * p.PB + c];
    for (int w = tid; w < p.P * parts; w += kScanThreads) {
        const int c = w % p.P, pt = w / p.P;
        const uint32_t ta = pt * per, tb = ta + per < nt ? ta + per : nt;
        uint32_t acc = 0;
#pragma unroll 8
        for (uint32_t t = ta; t < tb; ++t) acc += base[(size_t)t * p.PB + c];
        part[pt * p.PB + c] = acc;
    }
    __syncthreads();
    for (int c = tid; c < p.P; c += kScanThreads) {
        uint32_t s = 0;
        for (int pt = 0; pt < parts; ++pt) { const uint32_t v = part[pt * p.PB + c]; part[pt * p.PB + c] = s; s += v; }
    }
    __syncthreads();
    // zone < 3: the reference returns {{}, cleaned} before looking at any patch (recursive_patchwork.cpp:339-341)
    const uint32_t n_dropped = tot[p.P + 2], n_outside = tot[p.P + 1], n_ego = tot[p.P + 3];
    const bool zone_too_small = (nf - n_ego - n_dropped - n_outside) < 3u;
    // exclusive prefix over the patch classes, block-wide. Bucket positions: the block first ranks its patches
    // inside every size bucket in LDS, then reserves one range per bucket with a single global atomic (thousands of
    // returning atomics on a handful of counters cost tens of microseconds; the order inside a bucket is free).
    uint32_t running = 0, npatches = 0;
    for (int b = tid; b < kBuckets; b += kScanThreads) bcount[b] = 0;
    __syncthreads();
    for (int c0 = 0; c0 < p.P; c0 += kScanThreads) {
        const int c = c0 + tid;
        const uint32_t v = c < p.P ? tot[c] : 0u;
        uint32_t total, nz_total;
        const uint32_t ex = scan_block_excl_scan(v, red, total);
        (void)scan_block_excl_scan(v > 0 ? 1u : 0u, red, nz_total);
        if (c < p.P) {
            const uint32_t begin = running + ex;
            tot[c] = begin;
            p.patch_begin[(size_t)f * (p.P + 1) + c] = begin;
            PatchDesc d; d.begin = fo + begin; d.n = v;
            p.patches[(size_t)f * p.P + c] = d;
            // (also when zone < 3: patch_fit then only writes label 0)
            lrank[c] = v > 0 ? ((uint32_t)size_bucket(v) << 16) | atomicAdd(&bcount[size_bucket(v)], 1u) : 0xffffffffu;
        }
        running += total; npatches += nz_total;
    }
    __syncthreads();
    for (int b = tid; b < kBuckets; b += kScanThreads) { const uint32_t n = bcount[b]; bcount[b] = n ? atomicAdd(&p.bucket_count[b], n) : 0u; }
    __syncthreads();
    for (int c = tid; c < p.P; c += kScanThreads) {
        const uint32_t lr = lrank[c];
        if (lr != 0xffffffffu) {
            const uint32_t b = lr >> 16;
            p.bucket_list[(size_t)b * p.list_stride + bcount[b] + (lr & 0xffffu)] = f * (uint32_t)p.P + (uint32_t)c;
        }
    }
    // counts -> destinations: running offsets per class over the tiles, in tile (= input) order
    for (int w = tid; w < p.P * parts; w += kScanThreads) {
        const int c = w % p.P, pt = w / p.P;
        const uint32_t ta = pt * per, tb = ta + per < nt ? ta + per : nt;
        uint32_t off = tot[c] + part[pt * p.PB + c];
#pragma unroll 8
        for (uint32_t t = ta; t < tb; ++t) {
            uint32_t* cell = base + (size_t)t * p.PB + c;
            const uint32_t v = *cell;
            *cell = off;
            off += v;
        }
    }
    if (tid == 0) {
        p.patch_begin[(size_t)f * (p.P + 1) + p.P] = running;
        rpw_frame_stats st;
        st.n_in = nf - n_ego;                               // the fused cloud (everything, without a fusion step)
        st.n_clean = st.n_in - n_dropped;
        st.n_zone = st.n_clean - n_outside;
        st.n_ground = 0; st.n_patches = zone_too_small ? 0u : npatches; st.n_nodes = 0; st.n_splits = 0; st.max_depth = 0;
        st.total_iterations = 0; st.chain_addends = 0;
        p.stats[f] = st;
    }
}

// ------------------------------------------------------------------------------------------
// small pieces shared by the plane-fit kernel's passes
// ------------------------------------------------------------------------------------------
constexpr int kLdw  = 68;              // sweep ring row stride in floats: 64 + 4 keeps the 6 chain lanes on distinct banks
constexpr int kSlotFloats = 6 * kLdw;  // one ring slot: 64 addends of up to 6 rows

#define RPW_CHAIN4(ACC, Q) { ACC += (Q).x; ACC += (Q).y; ACC += (Q).z; ACC += (Q).w; }

__device__ __forceinline__ float plane_dist(float x, float y, float z, float cx_, float cy_, float cz_, float nx, float ny, float nz)
{
    const float e0 = (x - cx_) * nx;
    const float e1 = (y - cy_) * ny;
    const float e2 = (z - cz_) * nz;
    return fabs_(e0 + (e1 + e2));
}

struct PlaneD { float cx, cy, cz, nx, ny, nz, residual; };

// monotone key: ascending unsigned order == ascending float order
__device__ __forceinline__ uint32_t float_key(float v) { const uint32_t u = __float_as_uint(v); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float key_float(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

struct PatchAcc { uint32_t nodes, splits, maxdepth, ground; unsigned long long iters; };

__device__ __forceinline__ void emit_node(const DevParams& p, int tid, int frame, int patch, int depth, uint32_t start, uint32_t len,
                                          int kind, int iters, uint32_t ninl, int axis, float sval, const PlaneD* pl)
{
    if (!p.trace || tid != 0) return;
    const uint32_t slot = atomicAdd(p.node_count, 1u);
    if (slot >= p.nodes_cap) return;
    rpw_node nd;
    nd.frame = frame; nd.patch = patch; nd.depth = depth; nd.start = start; nd.len = len; nd.kind = kind;
    nd.iterations = iters; nd.n_inliers = ninl; nd.split_axis = axis; nd.split_value = sval;
    nd.centroid[0] = pl ? pl->cx : 0.0f; nd.centroid[1] = pl ? pl->cy : 0.0f; nd.centroid[2] = pl ? pl->cz : 0.0f;
    nd.normal[0] = pl ? pl->nx : 0.0f; nd.normal[1] = pl ? pl->ny : 0.0f; nd.normal[2] = pl ? pl->nz : 0.0f;
    nd.residual = pl ? pl->residual : 0.0f;
    p.nodes[slot] = nd;
}

// ==========================================================================================
// K3 patch_fit1: ONE WAVE per (frame, patch), nothing shared with any other wave.
//
// (Round 1's kernel paired a producer wave with a consumer wave per patch. It was bound by its producer: turning 64
// points into compacted addends cost it ~170 instructions per ring slot -- a lone wave issues one every ~5 cycles --
// while the chain needs 64 x 8.2 = 525 cycles, and every slot cost a rendezvous on top: 64.8 k frames/s. DESIGN.md
// section 4 keeps its measurements.) Here the work is cut differently:
//   * a MASK PASS (data-parallel, no chain) evaluates the inlier predicate of every slot of the node, keeps the
//     mask history byte and writes the slot numbers of the inliers, in slot order, into an index list;
//   * a CHAIN SWEEP walks that list 64 inliers at a time: gather x, y, z, make the K addends, transpose them
//     through a two-slot LDS ring, and run the K sequential sums (accumulator k in lane k) -- all in ONE
//     instruction stream. A dependent v_add_f32 issues every ~8 cycles and a lone wave issues every 4, so the
//     ~50 instructions that feed the next slot ride in the chain's shadow: the chain runs at its own speed and
//     no barrier, flag or second wave is involved.
// Per refit iteration: covariance sweep over the inliers (6 rows) -> eigenvector -> mask pass over the node ->
// centroid sweep over the new inliers (3 rows), or, when the mask did not change, the residual sweep (1 row).
// Everything else (early-outs, order statistics, partition, labels) is the same arithmetic as above,
// executed by 64 lanes. Sums are accumulated in exactly the reference's order (recursive_patchwork.cpp:98-104,
// 383-387, 241-248; point_cloud_processor.cpp:63-67, 78-83).
// ==========================================================================================
constexpr int kW1Ring = 3;           // ring slots: slot u lives in ring slot u % 3, a compile-time constant inside the loop unrolled by three

constexpr int kPassArgWords = 24;
struct Fit1Shared {
    __attribute__((aligned(16))) float ring[kRingFloats];              // sweeps: kW1Ring * kSlotFloats floats; fused pass: kFSlots * kFSlotFloats
    uint32_t hist[256];
    uint32_t stk[3 * kStackDepth];
    __attribute__((aligned(16))) uint32_t args[kPassArgWords];         // arguments of the server forms' pass functions (PassArgs below)
};
static_assert(kW1Ring * kSlotFloats <= kRingFloats, "the sweep ring must fit");

struct W1 {
    const float *sx, *sy, *sz;         // patch-sorted SoA
    uint32_t* il;                      // inlier index list of the current node (node-relative slot numbers, ascending)
    Fit1Shared* sh;
    uint32_t base;                     // absolute slot of the patch's first point
    int lane;
    unsigned long long addends;        // chain steps so far (statistics)
};

struct P4 { float x, y, z; };

// Ordering INSIDE one wave (a lane reads what another lane of the same wave wrote, in LDS or in global memory through the
// CU's L1): earlier accesses complete before later ones start. No s_barrier: the waves of a workgroup run independent
// control flow (a server wave and its patch owners).
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }

typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) uint32_t lds_word;

// The node a pass works on, as plain pointers to its first slot (by value: the passes are real functions, not inlined
// copies -- six inlined sweeps made the compiler hoist six sets of loop invariants over the whole node loop, 253 VGPRs).
// (explicit global address space: a plain pointer that arrives as a function argument is "generic", and generic
// accesses compile to flat_ instructions that count against the LDS counter as well)
typedef __attribute__((address_space(1))) const float g_cf32;
typedef __attribute__((address_space(1))) uint32_t g_u32;
typedef __attribute__((address_space(1))) unsigned long long g_u64;
// Inlier masks are BIT PLANES: one 64-bit word per chunk of 64 node slots (bit l = slot 64 c + l), three planes that rotate
// -- a pass reads the current mask and the one before it (two words per chunk) and writes the new one into the third. The masks
// are temporaries of the node being fitted, so every node of a patch uses the patch's words from their start.
// (Rounds 1-2 kept a history BYTE per slot: 1 B read + 1 B written per slot and refit, where this is 0.375 B.)
struct NodeRef {
    g_cf32 *x, *y, *z;                 // patch-sorted SoA, at the node's first slot
    g_u64 *mnew, *mcur, *mprev;        // mask planes: written by the pass / the current mask / the mask before it
    g_u32* il;                         // inlier index list (node-relative slot numbers, ascending)
    uint32_t n;                        // slots in the node
};
__device__ __forceinline__ float ldg_f32(g_cf32* base, uint32_t slot)
{
    return *reinterpret_cast<g_cf32*>(reinterpret_cast<__attribute__((address_space(1))) const char*>(base) + slot * 4u);
}

// Arguments of a real function arrive in vector registers; everything a pass gets is wave-uniform, and saying so
// (v_readfirstlane) turns its loops into scalar loops and its loads into the SGPR-base + 32-bit-offset form.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ float uni(float v) { return __uint_as_float(uni(__float_as_uint(v))); }
template <class T> __device__ __forceinline__ T* uni_ptr(T* p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned long long u = ((unsigned long long)uni((uint32_t)(v >> 32)) << 32) | (unsigned long long)uni((uint32_t)v);
    return (T*)u;
}
__device__ __forceinline__ NodeRef uni(NodeRef nd)
{
    nd.x = uni_ptr(nd.x); nd.y = uni_ptr(nd.y); nd.z = uni_ptr(nd.z); nd.mnew = uni_ptr(nd.mnew); nd.mcur = uni_ptr(nd.mcur); nd.mprev = uni_ptr(nd.mprev);
    nd.il = uni_ptr(nd.il); nd.n = uni(nd.n);
    return nd;
}

// slot i of the node, raw (i < n)
__device__ __forceinline__ P4 w1_load(const NodeRef& nd, uint32_t i)
{
    P4 r;
    r.x = ldg_f32(nd.x, i); r.y = ldg_f32(nd.y, i); r.z = ldg_f32(nd.z, i);
    return r;
}
// word c of a mask plane, the same for every lane (one request; the value is wave-uniform)
__device__ __forceinline__ unsigned long long mask_word(g_u64* plane, uint32_t c)
{
    const unsigned long long w = plane[c];
    return ((unsigned long long)uni((uint32_t)(w >> 32)) << 32) | (unsigned long long)uni((uint32_t)w);
}

// MASK PASS over the node, kMaskOlder: the index list of the mask one iteration back (plane `mprev`), inliers in slot order --
// for the residual of a period-2 fit that ends on the older mask. (The seed and plane predicates run in the fused pass below.)
//   kMaskSeed   inlier = z < prm.thr
//   kMaskPlane  inlier = |(p - c) . n| < prm.thr          flags: changed / differs from the mask before
//   kMaskOlder  inlier = the bit of the mask one iteration back
enum { kMaskSeed = 0, kMaskPlane = 1, kMaskOlder = 2 };
struct MaskPrm { float cx, cy, cz, nx, ny, nz, thr; };
struct MaskOut { uint32_t count, changed, differs_prev; };

struct yes { static constexpr bool value = true; };
struct no { static constexpr bool value = false; };

// one entry of the index list, 2 or 4 bytes wide (IW): nodes of up to 65536 slots use 16-bit entries
template <int IW> __device__ __forceinline__ void il_store(__attribute__((address_space(1))) char* il, uint32_t pos, uint32_t v)
{
    if (IW == 2) *reinterpret_cast<__attribute__((address_space(1))) uint16_t*>(il + pos * 2u) = (uint16_t)v;
    else         *reinterpret_cast<g_u32*>(il + pos * 4u) = v;
}
template <int IW> __device__ __forceinline__ uint32_t il_load(__attribute__((address_space(1))) const char* il, uint32_t pos)
{
    if (IW == 2) return (uint32_t)*reinterpret_cast<__attribute__((address_space(1))) const uint16_t*>(il + pos * 2u);
    return *reinterpret_cast<const g_u32*>(il + pos * 4u);
}

template <int IW>
__device__ __forceinline__ MaskOut w1_older_list(const NodeRef& nd, int lane)
{
    typedef __attribute__((address_space(1))) char g_char;
    g_char* il = reinterpret_cast<g_char*>(nd.il);
    const uint32_t n = nd.n, nchunks = (n + kWave - 1) / kWave;
    uint32_t cursor = 0;
    for (uint32_t c = 0; c < nchunks; ++c) {
        const uint32_t i = c * kWave + lane;
        unsigned long long m = mask_word(nd.mprev, c);
        if ((c + 1u) * kWave > n) m &= (1ull << (n & 63u)) - 1ull;      // the partial chunk: bits past the node are not slots
        const bool in = (m >> lane) & 1ull;
        if (in) il_store<IW>(il, cursor + (uint32_t)mbcnt64(m), i);
        cursor += (uint32_t)__popcll(m);
    }
    MaskOut o; o.count = cursor; o.changed = 0u; o.differs_prev = 0u;
    return o;
}

__device__ __attribute__((noinline)) MaskOut w1_mask_pass(NodeRef nd_in, int mode, MaskPrm prm)
{
    const int lane = threadIdx.x & 63;
    const NodeRef nd = uni(nd_in);
    (void)mode; (void)prm;
    if (nd.n <= 65536u) return w1_older_list<2>(nd, lane);
    return w1_older_list<4>(nd, lane);
}

// FUSED PASS = mask pass + centroid chain. The inliers of the NEW mask are found chunk by chunk, and their x, y, z go
// straight into an LDS ring in slot order, where lanes 0..2 add them up sequentially (accumulator k in lane k, one
// ds_read_b128 per four addends): the centroid sums of the new mask (point_cloud_processor.cpp:63-67) come out of the
// same pass over the node that decides the mask, and the points are not read a second time for them. The chain runs one
// chunk behind the predicate: while chunk c is evaluated (its loads were issued four chunks earlier), the complete
// groups of four addends that the chunks before it left in the ring are added; their first reads are issued before the
// chunk's arithmetic and return under it.
// fused ring: kFSlots slots of 64 addends x 3 rows (row stride kLdw, like a sweep slot); addend p lives in slot (p / 64) % kFSlots,
// column p % 64. The lanes that hold no inlier store into a dump row of their own behind the slots (one entry per lane:
// seven lanes per entry of a four-entry dump made every store a seven-way bank conflict).
constexpr int kFSlots = 8;
constexpr int kFSlotFloats = 3 * kLdw;
constexpr int kFDump = kFSlots * kFSlotFloats;
__device__ __forceinline__ uint32_t fused_ring_pos(uint32_t p) { return ((p >> 6) & (uint32_t)(kFSlots - 1)) * (uint32_t)kFSlotFloats + (p & 63u); }
struct FusedOut { uint32_t count, changed, differs_prev; float sx, sy, sz; };

// The centroid chain of a fused pass: whole slots of 64 addends, added with the sweeps' statically scheduled routine (one
// ds_read_b128 per four dependent adds, half a slot of reads in flight). A dynamic loop over groups of four addends cost
// ~21 cycles per add; this costs ~9.
struct FusedChain {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4f lds_f4;
    float acc = 0.0f;                  // Zero(), then += in slot order (lane k < 3: coordinate k)
    uint32_t cs = 0;                   // slots added so far
    bool have_a = false;               // A holds the first half of slot cs
    v4f A[8], B[8];
    __device__ __forceinline__ void reserve(uint32_t) {}           // (the server form waits for ring space here)
    __device__ __forceinline__ float result(int k) const { return __shfl(acc, k); }
    // add the slots [cs, avail): all of them are complete in the ring
    __device__ __forceinline__ void consume(const lds_f32* rowp, uint32_t avail)
    {
        while (cs < avail) {
            const lds_f32* sp = rowp + (cs & (uint32_t)(kFSlots - 1)) * (uint32_t)kFSlotFloats;
            if (!have_a) {
#pragma unroll
                for (int q = 0; q < 8; ++q) A[q] = *reinterpret_cast<const lds_f4*>(sp + 4 * q);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { B[q] = *reinterpret_cast<const lds_f4*>(sp + 32 + 4 * q); RPW_CHAIN4(acc, A[q]) __builtin_amdgcn_sched_barrier(0); }
            ++cs;
            if (cs < avail) {
                const lds_f32* np = rowp + (cs & (uint32_t)(kFSlots - 1)) * (uint32_t)kFSlotFloats;
#pragma unroll
                for (int q = 0; q < 8; ++q) { A[q] = *reinterpret_cast<const lds_f4*>(np + 4 * q); RPW_CHAIN4(acc, B[q]) __builtin_amdgcn_sched_barrier(0); }
                have_a = true;
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) { RPW_CHAIN4(acc, B[q]) }
                have_a = false;
            }
        }
    }
    // the end of the pass: `cursor` addends were written; the last slot is padded with -0.0f (the exact identity) and added
    __device__ __forceinline__ void finish(lds_f32* ring, const lds_f32* rowp, uint32_t cursor, int lane)
    {
        consume(rowp, cursor >> 6);
        if (cursor & 63u) {
            if ((uint32_t)lane >= (cursor & 63u)) {
                lds_f32* w = ring + fused_ring_pos((cursor & ~63u) + (uint32_t)lane);
                w[0] = -0.0f; w[kLdw] = -0.0f; w[2 * kLdw] = -0.0f;
            }
            have_a = false;                                        // (a prefetched half of this slot would predate the padding)
            consume(rowp, (cursor >> 6) + 1u);
        }
    }
};
static_assert(kFDump + 2 * kLdw + kWave <= kRingFloats, "the fused ring and its dump rows must fit");

template <int MODE, int IW, class CH>
__device__ __forceinline__ FusedOut w1_fused_loop(const NodeRef& nd, const MaskPrm& prm, lds_f32* ring, int lane, CH& ch)
{
    typedef __attribute__((address_space(1))) char g_char;
    g_char* il = reinterpret_cast<g_char*>(nd.il);
    const uint32_t n = nd.n;
    const uint32_t nfull = n / kWave;
    uint32_t cursor = 0;                                           // inliers so far = addends written
    bool changed = false, differs_prev = false;                    // wave-uniform: whole mask words are compared
    const int row = lane < 3 ? lane : 2;
    const lds_f32* rowp = ring + row * kLdw;
    // one chunk: predicate -> the new mask word (a ballot), compared as a whole with the chunk's word of the current mask
    // (wc) and of the mask before it (wp); list entries and centroid addends of the inliers
    auto step = [&](const P4& r, uint32_t c, unsigned long long wc, unsigned long long wp, auto full, auto cmp) __attribute__((always_inline)) -> unsigned long long {
        constexpr bool FULL = decltype(full)::value;
        constexpr bool CMP = decltype(cmp)::value;                 // compare the new word with wc / wp here (the rounds of four do it per lane)
        const uint32_t i = c * kWave + lane;
        const bool ok = FULL || i < n;
        bool in;
        if (MODE == kMaskPlane) {
            const float d = plane_dist(r.x, r.y, r.z, prm.cx, prm.cy, prm.cz, prm.nx, prm.ny, prm.nz);
            in = d < prm.thr;
        } else {
            in = r.z < prm.thr;
        }
        in = in && ok;
        const unsigned long long m = ballot64(in);
        if (MODE == kMaskPlane && CMP) {
            const unsigned long long valid = FULL ? ~0ull : ((1ull << (n & 63u)) - 1ull);
            changed |= m != (wc & valid);
            differs_prev |= m != (wp & valid);
        }
        const uint32_t p = cursor + (uint32_t)mbcnt64(m);
        // (selections by bit mask, not by ?: -- two selects on one condition were turned into an if / else with a store
        // in each arm, i.e. into branches inside the chunk)
        const uint32_t sel = 0u - (uint32_t)in;
        // index list (for the single-wave form's covariance sweep; IW == 0, the server forms: none, their sweeps read the mask):
        // an inlier to its place, everybody else into the last entry
        if (IW) {
            const uint32_t ip = (p & sel) | ((n - 1u) & ~sel);
            if (FULL) il_store<(IW ? IW : 4)>(il, ip, i);
            else if (ok) il_store<(IW ? IW : 4)>(il, ip, i);
        }
        // centroid addends: an inlier to its place in the ring, everybody else into the dump entries
        const uint32_t lp = (fused_ring_pos(p) & sel) | (((uint32_t)kFDump + (uint32_t)lane) & ~sel);
        lds_f32* w = ring + lp;
        w[0] = r.x; w[kLdw] = r.y; w[2 * kLdw] = r.z;
        cursor += (uint32_t)__popcll(m);
        return m;
    };
    // Four full chunks per round: the reads of the groups the earlier rounds completed, then the four chunks' arithmetic
    // as ONE basic block (so that the waits for their loads count precisely), then the adds of those groups.
    // The first round is PEELED out of the loop: the wait counts inside a loop must hold for the way it is entered as
    // well, and entered straight from the four back-to-back prologue loads they came out as vmcnt(7) where the steady
    // state allows vmcnt(15) -- every chunk then waited for loads issued one chunk earlier.
    // The mask words of a round's four chunks are loaded a round ahead like the points: lane j holds the words of chunk
    // c + (j & 3) and, at the end of the round, the new word of that chunk (the value it stores): the comparisons with the
    // current mask and the one before it are two 64-bit vector compares per round (sixteen lanes repeat each), where copying
    // the eight old words to scalars cost sixteen v_readlane and as many scalar compares.
    uint32_t c = 0;
    if (nfull >= 4) {
        P4 r0 = w1_load(nd, 0 * kWave + lane), r1 = w1_load(nd, 1 * kWave + lane), r2 = w1_load(nd, 2 * kWave + lane), r3 = w1_load(nd, 3 * kWave + lane);
        const uint32_t lastfull = (nfull - 1u) * kWave;
        const uint32_t lastword = nfull - 1u;
        auto words_of = [&](g_u64* plane, uint32_t cb) -> unsigned long long {
            const uint32_t w = cb + ((uint32_t)lane & 3u);
            return plane[w < lastword ? w : lastword];
        };
        unsigned long long WC = 0, WP = 0, chg = 0ull, dif = 0ull;
        if (MODE == kMaskPlane) { WC = words_of(nd.mcur, 0); WP = words_of(nd.mprev, 0); }
        auto round = [&]() __attribute__((always_inline)) {
            // (the reloads clamp to the last full chunk: loads past it would be wasted, never wrong)
            const uint32_t b4 = (c + 4) * kWave, b5 = b4 + kWave, b6 = b5 + kWave, b7 = b6 + kWave;
            ch.reserve(cursor + 4u * kWave);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long m0 = step(r0, c, 0ull, 0ull, yes(), no());     r0 = w1_load(nd, (b4 < lastfull ? b4 : lastfull) + lane); __builtin_amdgcn_sched_barrier(0);
            const unsigned long long m1 = step(r1, c + 1, 0ull, 0ull, yes(), no()); r1 = w1_load(nd, (b5 < lastfull ? b5 : lastfull) + lane); __builtin_amdgcn_sched_barrier(0);
            const unsigned long long m2 = step(r2, c + 2, 0ull, 0ull, yes(), no()); r2 = w1_load(nd, (b6 < lastfull ? b6 : lastfull) + lane); __builtin_amdgcn_sched_barrier(0);
            const unsigned long long m3 = step(r3, c + 3, 0ull, 0ull, yes(), no()); r3 = w1_load(nd, (b7 < lastfull ? b7 : lastfull) + lane); __builtin_amdgcn_sched_barrier(0);
            // the round's four new mask words in ONE store: lane j writes the word of chunk c + (j & 3) (sixteen lanes per word)
            {
                const uint32_t k = (uint32_t)lane & 3u;
                const unsigned long long lo = (k & 1u) ? m1 : m0, hi = (k & 1u) ? m3 : m2;
                const unsigned long long mine = (k & 2u) ? hi : lo;
                nd.mnew[c + k] = mine;
                if (MODE == kMaskPlane) {
                    chg |= ballot64(mine != WC);
                    dif |= ballot64(mine != WP);
                    WC = words_of(nd.mcur, c + 4); WP = words_of(nd.mprev, c + 4);      // the next round's, needed at its end
                }
            }
            ch.consume(rowp, cursor >> 6);                            // the slots this round completed
            c += 4;
        };
        round();
        while (c + 4 <= nfull) round();
        changed |= chg != 0ull;
        differs_prev |= dif != 0ull;
    }
    for (; c < nfull; ++c) {                                       // up to three full chunks left
        const P4 r = w1_load(nd, c * kWave + lane);
        unsigned long long wc = 0, wp = 0;
        if (MODE == kMaskPlane) { wc = mask_word(nd.mcur, c); wp = mask_word(nd.mprev, c); }
        ch.reserve(cursor + kWave);
        nd.mnew[c] = step(r, c, wc, wp, yes(), yes());               // (every lane the same word to the same address)
        ch.consume(rowp, cursor >> 6);
    }
    if (nfull * kWave < n) {                                       // the partial chunk
        const uint32_t i = nfull * kWave + lane;
        const P4 r = w1_load(nd, i < n ? i : n - 1u);
        unsigned long long wc = 0, wp = 0;
        if (MODE == kMaskPlane) { wc = mask_word(nd.mcur, nfull); wp = mask_word(nd.mprev, nfull); }
        ch.reserve(cursor + kWave);
        nd.mnew[nfull] = step(r, nfull, wc, wp, no(), yes());
        ch.consume(rowp, cursor >> 6);
    }
    ch.finish(ring, rowp, cursor, lane);
    FusedOut o;
    o.count = cursor;
    o.changed = changed ? 1u : 0u;
    o.differs_prev = differs_prev ? 1u : 0u;
    o.sx = ch.result(0); o.sy = ch.result(1); o.sz = ch.result(2);
    return o;
}

__device__ __attribute__((noinline)) FusedOut w1_fused_pass(NodeRef nd_in, int mode, MaskPrm prm, lds_f32* ring)
{
    const int lane = threadIdx.x & 63;
    const NodeRef nd = uni(nd_in);
    prm.cx = uni(prm.cx); prm.cy = uni(prm.cy); prm.cz = uni(prm.cz); prm.nx = uni(prm.nx); prm.ny = uni(prm.ny); prm.nz = uni(prm.nz);
    prm.thr = uni(prm.thr);
    ring = (lds_f32*)(uintptr_t)uni((uint32_t)(uintptr_t)ring);
    const bool narrow = nd.n <= 65536u;
    FusedChain ch;
    if (uni((uint32_t)mode) == (uint32_t)kMaskPlane)
        return narrow ? w1_fused_loop<kMaskPlane, 2>(nd, prm, ring, lane, ch) : w1_fused_loop<kMaskPlane, 4>(nd, prm, ring, lane, ch);
    return narrow ? w1_fused_loop<kMaskSeed, 2>(nd, prm, ring, lane, ch) : w1_fused_loop<kMaskSeed, 4>(nd, prm, ring, lane, ch);
}

struct X3 { float x, y, z; };

// CHAIN SWEEP: K sequential sums over `count` elements of the node, element j being slot il[j] (indexed) or slot j
// (every slot of the node). sums[k] = ((a_0 + a_1) + ...) in element order, starting from +0, exactly as a scalar loop
// accumulates them; accumulator k lives in lane k. What an element's addends are:
//   kSweepXYZ     x, y, z                                   centroid sums (point_cloud_processor.cpp:63-67)
//   kSweepCov     products of (p - c), lower triangle       covariance sums (point_cloud_processor.cpp:78-83); prm = c
//   kSweepDist    |(p - c) . n|                             residual sum (recursive_patchwork.cpp:98-104); prm = c, n
//   kSweepRange   sqrtf(x*x + y*y)                          mean distance (recursive_patchwork.cpp:383-387)
//   kSweepXY      x, y                                      centroid for the split variance (recursive_patchwork.cpp:239)
//   kSweepVarXY   (x - mx)^2, (y - my)^2                    split variances (recursive_patchwork.cpp:241-248); prm = mx, my
// The loop is ONE instruction stream: the 64 dependent adds of slot t (one per ~8 cycles) with the loads, arithmetic
// and LDS transposition of slot t + 1 issued in between.
enum { kSweepXYZ = 0, kSweepCov = 1, kSweepDist = 2, kSweepRange = 3, kSweepXY = 4, kSweepVarXY = 5 };
struct SweepPrm { float a, b, c, d, e, f; };
struct Sums6 { float v[6]; };

template <int MODE> struct SweepRows { static constexpr int K = MODE == kSweepCov ? 6 : (MODE == kSweepXYZ ? 3 : ((MODE == kSweepXY || MODE == kSweepVarXY) ? 2 : 1)); };

// One slot = 64 elements = 16 steps of {one ds_read_b128 of the row, four dependent adds}. A step's adds take ~33 cycles
// of latency but only five issue slots, so every step carries a few instructions of the NEXT slots' preparation; the
// scheduling barriers pin them there (left alone, the compiler gathers the LDS reads right in front of their use).
//   steps 0-7  (first half of slot t, second half's reads)   : addends of slot t + 1 -> ring
//   steps 8-15 (second half of slot t, reads of slot t + 1)   : point loads of slot t + 4, index load of slot t + 6
template <int MODE, int IW>        // IW: bytes per index list entry, 0 = the sweep runs over every slot of the node
__device__ __forceinline__ float w1_sweep_loop(const NodeRef& nd, uint32_t count, const SweepPrm& prm, lds_f32* ring, int lane, float& zmn, float& zmx)
{
    constexpr int K = SweepRows<MODE>::K;
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4f lds_f4;
    const uint32_t nslots = (count + kWave - 1) / kWave;
    typedef __attribute__((address_space(1))) const char g_cchar;
    g_cchar* il = reinterpret_cast<g_cchar*>(nd.il);
    const uint32_t last = count - 1u;
    // element j of slot u -> node slot (clamped: lanes past the end repeat the last element; their addends are replaced)
    auto index_of = [&](uint32_t u) -> uint32_t {
        const uint32_t j = u * kWave + lane, jc = j < last ? j : last;
        return IW ? il_load<(IW ? IW : 4)>(il, jc) : jc;
    };
    auto point_of = [&](uint32_t idx) -> X3 {
        X3 q; q.x = ldg_f32(nd.x, idx); q.y = ldg_f32(nd.y, idx);
        q.z = (MODE == kSweepXYZ || MODE == kSweepCov || MODE == kSweepDist || MODE == kSweepRange) ? ldg_f32(nd.z, idx) : 0.0f;
        return q;
    };
    // addends of one element, in two halves (so that they can sit in two different steps)
    float v[6];
    f2 dxy; float dz;
    X3 P1, P2, P0;
    auto make_a = [&](const X3& q) __attribute__((always_inline)) {
        if (MODE == kSweepCov) {
            // two-wide float arithmetic (v_pk_add_f32 / v_pk_mul_f32: each half is an ordinary IEEE binary32 operation)
            dxy = f2{q.x, q.y} - f2{prm.a, prm.b};
            dz = q.z - prm.c;
            const f2 a = dxy * f2{dxy.x, dxy.x};            // dx*dx, dy*dx
            v[0] = a.x; v[1] = a.y;                         // (0,0) (1,0)
        } else if (MODE == kSweepXYZ) { v[0] = q.x; v[1] = q.y; v[2] = q.z; }
        else if (MODE == kSweepDist)  { v[0] = plane_dist(q.x, q.y, q.z, prm.a, prm.b, prm.c, prm.d, prm.e, prm.f); }
        else if (MODE == kSweepRange) { v[0] = sqrt_rn(q.x * q.x + q.y * q.y); zmn = fminf(zmn, q.z); zmx = fmaxf(zmx, q.z); }      // (+ the node's z range: order-free)
        else if (MODE == kSweepXY)    { v[0] = q.x; v[1] = q.y; }
        else { const float dx = q.x - prm.a, dy = q.y - prm.b; v[0] = dx * dx; v[1] = dy * dy; }
    };
    auto make_b = [&]() __attribute__((always_inline)) {
        if (MODE == kSweepCov) {
            const f2 b = f2{dxy.y, dz} * f2{dxy.y, dxy.x};  // dy*dy, dz*dx
            const f2 c = f2{dz, dz} * f2{dxy.y, dz};        // dz*dy, dz*dz
            v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y; // (1,1) (2,0) (2,1) (2,2)
        }
    };
    // rows [k0, k1) of slot u into ring slot u & 1 (row k at k * kLdw, column = lane); columns past the end get -0.0f,
    // the exact additive identity (a select, not a branch: the loop body must stay ONE basic block, or the adds
    // are sunk out of their steps)
    auto put = [&](uint32_t u, int rs, int k0, int k1) __attribute__((always_inline)) {
        lds_f32* dst = ring + rs * kSlotFloats + lane;
        const bool live = u * kWave + lane < count;
#pragma unroll
        for (int k = k0; k < k1 && k < K; ++k) dst[k * kLdw] = live ? v[k] : -0.0f;
    };
    // Pipeline: the index list runs two slots ahead of the point loads, the points three slots ahead of their addends.
    // Three register sets each, used round robin and reloaded right after use (the loop is unrolled by three): nothing
    // is ever copied, so no instruction touches a value that has just been requested from memory.
    //   P[u % 3] holds the points of slot u; I[u % 3] the indices of slot u.
    // (prologue in the loop's own order -- indices first, then the points, slot 0 first -- so that the loop is entered
    // with the same kind of loads in flight as its back edge carries; entered with none, its first wait became vmcnt(0))
    const uint32_t j0 = index_of(0), j1 = index_of(1), j2 = index_of(2), j3 = index_of(3);
    uint32_t I1 = index_of(4), I2 = index_of(5), I0 = 0;
    {
        const X3 p0 = point_of(j0);
        X3 q1 = point_of(j1), q2 = point_of(j2), q3 = point_of(j3);
        make_a(p0); make_b(); put(0, 0, 0, 6);
        P1 = q1; P2 = q2; P0 = q3;
    }
    const int row = lane < K ? lane : K - 1;
    const lds_f32* rowp = ring + row * kLdw;
    float acc = 0.0f;
    v4f A[8], B[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) A[q] = *reinterpret_cast<const lds_f4*>(rowp + 4 * q);
#define RPW_W1_STEP_B(Q, FILL) { B[Q] = *reinterpret_cast<const lds_f4*>(cur + 4 * (Q)); RPW_CHAIN4(acc, A[Q]) FILL; __builtin_amdgcn_sched_barrier(0); }
#define RPW_W1_STEP_A(Q, FILL) { A[Q] = *reinterpret_cast<const lds_f4*>(nxt + 4 * (Q)); RPW_CHAIN4(acc, B[Q]) FILL; __builtin_amdgcn_sched_barrier(0); }
    // slot T: P_CUR holds slot T + 1 and is reloaded with slot T + 4 (indices I_USE); I_LOAD receives the indices of slot T + 6
#define RPW_W1_SLOT(T, RS, P_CUR, I_USE, I_LOAD) {                                                          \
        const lds_f32* cur = rowp + (RS) * kSlotFloats + 32;                                                \
        const lds_f32* nxt = rowp + (((RS) + 1) % kW1Ring) * kSlotFloats;                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        RPW_W1_STEP_B(0, make_a(P_CUR))                                                                     \
        RPW_W1_STEP_B(1, make_b())                                                                          \
        RPW_W1_STEP_B(2, put((T) + 1, ((RS) + 1) % kW1Ring, 0, 2))                                          \
        RPW_W1_STEP_B(3, put((T) + 1, ((RS) + 1) % kW1Ring, 2, 4))                                          \
        RPW_W1_STEP_B(4, put((T) + 1, ((RS) + 1) % kW1Ring, 4, 6))                                          \
        RPW_W1_STEP_B(5, (void)0)                                                                           \
        RPW_W1_STEP_B(6, (void)0)                                                                           \
        RPW_W1_STEP_B(7, (void)0)                                                                           \
        RPW_W1_STEP_A(0, (P_CUR = point_of(I_USE)))                                                         \
        RPW_W1_STEP_A(1, (I_LOAD = index_of((T) + 6)))                                                      \
        RPW_W1_STEP_A(2, (void)0)                                                                           \
        RPW_W1_STEP_A(3, (void)0)                                                                           \
        RPW_W1_STEP_A(4, (void)0)                                                                           \
        RPW_W1_STEP_A(5, (void)0)                                                                           \
        RPW_W1_STEP_A(6, (void)0)                                                                           \
        RPW_W1_STEP_A(7, (void)0)                                                                           \
    }
    // three slots per round; the first round is peeled out of the loop (as in the fused pass: the wait counts of a loop
    // must also hold for the way it is entered, and the prologue's back-to-back loads made its first wait a vmcnt(0))
    auto round = [&](uint32_t t) __attribute__((always_inline)) -> bool {
        RPW_W1_SLOT(t, 0, P1, I1, I0)
        if (t + 1 >= nslots) return false;
        RPW_W1_SLOT(t + 1, 1, P2, I2, I1)
        if (t + 2 >= nslots) return false;
        RPW_W1_SLOT(t + 2, 2, P0, I0, I2)
        return t + 3 < nslots;
    };
    if (round(0)) {
        uint32_t t = 3;
        while (round(t)) t += 3;
    }
#undef RPW_W1_SLOT
#undef RPW_W1_STEP_A
#undef RPW_W1_STEP_B
    return acc;
}

__device__ __attribute__((noinline)) Sums6 w1_chain_sweep(NodeRef nd_in, uint32_t count, int mode, int indexed, SweepPrm prm, lds_f32* ring)
{
    const int lane = threadIdx.x & 63;
    const NodeRef nd = uni(nd_in);
    count = uni(count);
    prm.a = uni(prm.a); prm.b = uni(prm.b); prm.c = uni(prm.c); prm.d = uni(prm.d); prm.e = uni(prm.e); prm.f = uni(prm.f);
    ring = (lds_f32*)(uintptr_t)uni((uint32_t)(uintptr_t)ring);
    Sums6 out;
#pragma unroll
    for (int k = 0; k < 6; ++k) out.v[k] = 0.0f;
    if (count == 0) return out;
    float acc;
    float zmn = 3.402823466e+38f, zmx = -3.402823466e+38f;      // kSweepRange also returns the z range of the elements (v[1], v[2])
    // (mode and indexed arrive in vector registers; the wave-uniform copy makes these scalar branches)
    const bool narrow = nd.n <= 65536u;                // the width the pass that wrote the list used
    const int m = __builtin_amdgcn_readfirstlane(mode);
    switch (m) {
    case kSweepCov:   acc = narrow ? w1_sweep_loop<kSweepCov, 2>(nd, count, prm, ring, lane, zmn, zmx) : w1_sweep_loop<kSweepCov, 4>(nd, count, prm, ring, lane, zmn, zmx); break;
    case kSweepXYZ:   acc = narrow ? w1_sweep_loop<kSweepXYZ, 2>(nd, count, prm, ring, lane, zmn, zmx) : w1_sweep_loop<kSweepXYZ, 4>(nd, count, prm, ring, lane, zmn, zmx); break;
    case kSweepDist:  acc = narrow ? w1_sweep_loop<kSweepDist, 2>(nd, count, prm, ring, lane, zmn, zmx) : w1_sweep_loop<kSweepDist, 4>(nd, count, prm, ring, lane, zmn, zmx); break;
    case kSweepRange: acc = w1_sweep_loop<kSweepRange, 0>(nd, count, prm, ring, lane, zmn, zmx); break;
    case kSweepXY:    acc = w1_sweep_loop<kSweepXY, 0>(nd, count, prm, ring, lane, zmn, zmx); break;
    default:          acc = w1_sweep_loop<kSweepVarXY, 0>(nd, count, prm, ring, lane, zmn, zmx); break;
    }
    (void)indexed;
#pragma unroll
    for (int k = 0; k < 6; ++k) out.v[k] = __shfl(acc, k);
    if (m == kSweepRange) { out.v[1] = wave_min(zmn); out.v[2] = wave_max(zmx); }
    return out;
}

// exact order statistic, single wave (see select_rank)
__device__ __attribute__((noinline)) float w1_select_rank(W1 cx, uint32_t s, uint32_t n, int axis, uint32_t k)
{
    const float* src = (axis == 0 ? cx.sx : (axis == 1 ? cx.sy : cx.sz)) + cx.base + s;
    uint32_t* hist = cx.sh->hist;
    uint32_t prefix = 0, pmask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int j = cx.lane; j < 256; j += kWave) hist[j] = 0;
        wave_sync();
        for (uint32_t i = cx.lane; i < n; i += kWave) {
            const uint32_t key = float_key(src[i]);
            if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        wave_sync();
        const uint32_t h0 = hist[4 * cx.lane], h1 = hist[4 * cx.lane + 1], h2 = hist[4 * cx.lane + 2], h3 = hist[4 * cx.lane + 3];
        const uint32_t tot = h0 + h1 + h2 + h3;
        const uint32_t incl = wave_incl_scan(tot);
        const uint32_t excl = incl - tot;
        const bool mine = (k >= excl) && (k < incl);
        const unsigned long long bm = ballot64(mine);
        const int owner = __ffsll((long long)bm) - 1;
        uint32_t bin = 0, before = 0;
        if (mine) {
            uint32_t c = excl;
            if (k < c + h0) { bin = 4 * cx.lane; before = c; }
            else { c += h0; if (k < c + h1) { bin = 4 * cx.lane + 1; before = c; }
            else { c += h1; if (k < c + h2) { bin = 4 * cx.lane + 2; before = c; }
            else { c += h2; bin = 4 * cx.lane + 3; before = c; } } }
        }
        bin = __shfl(bin, owner); before = __shfl(before, owner);
        prefix |= bin << shift; pmask |= 255u << shift;
        k -= before;
        wave_sync();
    }
    return key_float(prefix);
}

// libstdc++ partial_sort(3 lowest z) restated, single wave (see three_lowest)
struct Sel3 { uint32_t v[3]; };
__device__ __attribute__((noinline)) Sel3 w1_three_lowest(W1 cx, uint32_t s, uint32_t n)
{
    const float* zs = cx.sz + cx.base + s;
    uint32_t h0 = 0, h1 = 1, h2 = 2;
    float z0 = zs[0], z1 = zs[1], z2 = zs[2];
    auto adjust = [&](uint32_t vi, float vz) {
        int hole;
        if (z2 < z1) { h0 = h1; z0 = z1; hole = 1; } else { h0 = h2; z0 = z2; hole = 2; }
        if (z0 < vz) { if (hole == 1) { h1 = h0; z1 = z0; } else { h2 = h0; z2 = z0; } h0 = vi; z0 = vz; }
        else { if (hole == 1) { h1 = vi; z1 = vz; } else { h2 = vi; z2 = vz; } }
    };
    adjust(h0, z0);
    for (uint32_t b = 3; b < n; b += kWave) {
        const uint32_t i = b + cx.lane;
        const float zi = i < n ? zs[i] : 0.0f;
        unsigned long long done = 0ull;
        for (;;) {
            const bool want = i < n && zi < z0 && !((done >> cx.lane) & 1ull);
            const unsigned long long cand = ballot64(want);
            if (!cand) break;
            const int l = __ffsll((long long)cand) - 1;
            const uint32_t vi = (uint32_t)__shfl((int)i, l);
            const float vz = __shfl(zi, l);
            adjust(vi, vz);
            done |= (l == 63) ? ~0ull : ((1ull << (l + 1)) - 1ull);
        }
    }
    Sel3 sel; sel.v[0] = h0; sel.v[1] = h1; sel.v[2] = h2;
    return sel;
}

constexpr int kFit1Threads = kWave;

struct N3 { float x, y, z; };
// the Eigen 3.4.0 solver (rpw_math.h) as a real function: it is called once per refit, inlining it twice only costs registers
__device__ __attribute__((noinline)) N3 w1_normal(float a00, float a10, float a11, float a20, float a21, float a22)
{
    N3 r;
    smallest_eigenvector(a00, a10, a11, a20, a21, a22, r.x, r.y, r.z);
    return r;
}

// ==========================================================================================
// CHAIN-SERVER FORM (the default). In the forms above every dependent add is a whole vector instruction that keeps 3 or 6
// of 64 lanes busy, and a saturated SIMD issues one vector instruction per ~4.6 cycles whatever its lane mask
// (tools/microbench_exec.hip): with one chain per wave the chip spends most of its issue slots on those adds. Here a
// workgroup is one SERVER wave + kSrvFeeders FEEDER waves:
//   feeder f : owns one patch at a time (taken from the launch's work counter, largest size bucket first) and runs
//              everything of the forms above EXCEPT the dependent adds: it writes each slot of 64 addends into its ring,
//              publishes the slot count, and at the end of a pass collects the sums;
//   server   : lanes 6f .. 6f+5 hold the up-to-six accumulators of feeder f. One round = one slot of 64 addends for EVERY
//              feeder that has one ready -- the same 64 add instructions serve all of them. A lane with nothing to add
//              reads a slot of -0.0f (x + -0.0f == x exactly), so the round is branch-free.
// A feeder's sums are still ((a0 + a1) + a2) + ... in element order from +0: each ring is added by one lane per row, slot
// after slot. Hand-over is by LDS words only (no s_barrier: the eight waves run independent control flow):
//   feed[f] = {slots published (counts up for ever), that counter's value when the pass began, layout, rows}   feeder -> server
//   cons[f] = slots added, acc[f][row] = the accumulator after them                                      server -> feeder
// One wave's LDS operations are performed in order, so "addends, then the count" / "accumulator, then cons" need no fence.
// Waits are polls with s_sleep and a spin limit (a protocol error ends the launch with an error flag, not a hung GPU).
// ==========================================================================================
#ifndef RPW_LIST_MAX_SLOTS
#define RPW_LIST_MAX_SLOTS 0            // batch form: nodes of up to this many slots keep an index list (their passes hit in L2: the list costs no
#endif                                  // HBM bytes there and saves the mask expansion's instructions); 0 = every node expands its mask planes
constexpr uint32_t kListMaxSlots = RPW_LIST_MAX_SLOTS;
static_assert(kListMaxSlots <= 65536u, "nodes with a list in the batch form use 16-bit entries");
#ifndef RPW_SRV1_MASKED
#define RPW_SRV1_MASKED 0               // 1: the server-per-patch form takes the inliers from the mask planes too (no index list)
#endif
constexpr int kSrvFeeders = 7;
constexpr int kSrvSweepSlots = 4;        // ring slots of a sweep (6 rows of 64 each); the fused pass uses kFSlots slots of 3 rows
#ifndef RPW_SWEEP_DEPTH
#define RPW_SWEEP_DEPTH 4
#endif
constexpr int kSrvSweepDepth = RPW_SWEEP_DEPTH;      // slots a sweep's feeder loads ahead (points; their indices as far again)
constexpr uint32_t kSrvSpinLimit = 1u << 25;      // polls (>= ~200 cycles each: seconds) before a wait gives up
static_assert(kSrvSweepSlots * kSlotFloats <= kRingFloats, "the server form's sweep ring must fit");
static_assert(6 * kSrvFeeders <= kWave, "six server lanes per feeder");

struct SrvShared {
    __attribute__((aligned(16))) uint32_t feed[8][4];
    __attribute__((aligned(16))) float acc[8][8];
    __attribute__((aligned(16))) float zero[64];
    uint32_t cons[8];
    uint32_t done;                     // feeders that have left
    uint32_t error;                    // a wait ran into the spin limit
    uint32_t pad[2];
};
__device__ __forceinline__ uint32_t srv_layout(uint32_t stride, uint32_t slots) { return stride | ((slots - 1u) << 16); }

// what a feeder needs of the hand-over block (LDS pointers, wave-uniform)
struct SrvPort {
    lds_word* fw;                      // feed[f]
    lds_word* cw;                      // &cons[f]
    lds_f32* accp;                     // acc[f]
    lds_word* err;                     // &error
    uint32_t pub;                      // per lane: where a publish stores -- lane 0 feed[f][0], every other lane a dump word of its own
};                                     // (in the owner's histogram block, idle during a pass: no exec mask to set up and restore)
constexpr int kPubDumpAt = 128;        // words of Fit1Shared::hist behind the sweeps' index buffer
__device__ __forceinline__ SrvPort srv_port(SrvShared* sv, int f, lds_f32* ring)
{
    SrvPort pt;
    pt.fw = (lds_word*)sv->feed[f]; pt.cw = (lds_word*)&sv->cons[f]; pt.accp = (lds_f32*)sv->acc[f]; pt.err = (lds_word*)&sv->error;
    const int lane = threadIdx.x & 63;
    const uint32_t dump = (uint32_t)(uintptr_t)ring + (uint32_t)offsetof(Fit1Shared, hist) + (uint32_t)(kPubDumpAt + lane) * 4u;
    pt.pub = lane == 0 ? (uint32_t)(uintptr_t)pt.fw : dump;
    return pt;
}
__device__ __forceinline__ uint32_t srv_begin(const SrvPort& pt, uint32_t layout, uint32_t rows, int lane)
{
    const uint32_t start = uni(*(volatile lds_word*)pt.fw);
    if (lane == 0) { pt.fw[1] = start; pt.fw[2] = layout; pt.fw[3] = rows; }
    asm volatile("" ::: "memory");
    return start;
}
// (compiler barriers: a volatile access orders only against other volatile accesses, so without them the compiler
// could sink a slot's plain LDS stores below the count that announces them, or lift stores into a ring slot above the wait for
// it; the hardware performs one wave's LDS operations in order, so no instruction is needed)
__device__ __forceinline__ void srv_publish(const SrvPort& pt, uint32_t abs_slots, int lane)
{
    asm volatile("" ::: "memory");
    *(volatile lds_word*)(uintptr_t)pt.pub = abs_slots;
    asm volatile("" ::: "memory");
}
// until the server has added `need` slots (absolute count); `seen` caches the last value read
__device__ __forceinline__ void srv_wait(const SrvPort& pt, uint32_t need, uint32_t& seen)
{
    uint32_t spins = 0;
    while ((int32_t)(seen - need) < 0) {
        seen = uni(*(volatile lds_word*)pt.cw);
        const uint32_t broken = uni(*(volatile lds_word*)pt.err);
        if ((int32_t)(seen - need) >= 0 || broken) break;          // (after one time-out nobody in the workgroup waits again)
        __builtin_amdgcn_s_sleep(0);
        if (++spins > kSrvSpinLimit) { *(volatile lds_word*)pt.err = 1u; break; }
    }
    asm volatile("" ::: "memory");
}

// The server wave. Round k adds slot `cons` of every ready ring; the hand-over words for round k + 1 are read at the start
// of round k and turned into addresses in the middle of it, so that the first half of the next slots is requested under
// the second half of this round's adds: the 64 dependent adds of consecutive rounds follow each other without a gap.
// The wave runs at raised priority: its adds are the critical path of up to seven patches, and an add issues only every
// ~8 cycles -- the feeders' instructions fit in between.
__device__ __attribute__((noinline)) void srv_loop(SrvShared* sv, Fit1Shared* shs)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4f lds_f4;
    typedef __attribute__((address_space(3))) v4u lds_u4;
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63;
    const int grp = lane / 6, row = lane - 6 * grp;
    const bool valid = grp < kSrvFeeders;
    const int g = valid ? grp : 0;
    const lds_f32* ringrow = (lds_f32*)shs[g].ring + row * kLdw;
    const lds_f32* zero = (lds_f32*)sv->zero;
    volatile lds_u4* fw = (volatile lds_u4*)sv->feed[g];
    volatile lds_word* donep = (volatile lds_word*)&sv->done;
    lds_f32* accw = (lds_f32*)&sv->acc[g][row];
    lds_word* consw = (lds_word*)&sv->cons[g];
    uint32_t cons = 0;
    float acc = 0.0f;
    v4f A[8], B[8];
    // what a round needs to know about its slot: is there one (ready), does this lane take part (act), does a pass begin
    bool ready = false, act = false, first = false;
    const lds_f32* src = zero;
    auto look = [&](const v4u& st, uint32_t c) __attribute__((always_inline)) {      // the slot of the round that starts with `c` slots added
        ready = valid && (int32_t)(st.x - c) > 0;
        // (all four words are used: a dead fourth word became the compiler's scratch register, and writing it meant
        // waiting for the read at once)
        const uint32_t stride = st.z & 0xffffu, nsm = st.z >> 16, rows = st.w;
        act = ready && (uint32_t)row < rows;
        first = act && c == st.y;
        src = act ? ringrow + ((c - st.y) & nsm) * stride : zero;
    };
    for (;;) {
        // ---- nothing in flight: wait for a slot (or for the feeders to leave) ----
        // (No limit on the server's own polls: every feeder's waits are bounded, so every feeder leaves, and the server leaves
        // behind the last of them. A limit here could only fire on feeders that are legitimately busy elsewhere -- the bounding box
        // or the radix select of a very large patch.)
        for (;;) {
            const v4u st = *fw;
            const uint32_t done = *donep;
            look(st, cons);
            if (ballot64(ready) != 0ull) break;
            if (uni(done) >= (uint32_t)kSrvFeeders) return;
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) A[q] = *reinterpret_cast<const lds_f4*>(src + 4 * q);
        // ---- rounds back to back while there is work ----
        for (;;) {
            acc = first ? 0.0f : acc;                              // the first slot of a pass: Zero(), then +=
            const bool ready_k = ready, act_k = act;
            const lds_f32* src_k = src;
            const v4u nst = *fw;                                   // hand-over words for the next round, requested now
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; ++q) { B[q] = *reinterpret_cast<const lds_f4*>(src_k + 32 + 4 * q); RPW_CHAIN4(acc, A[q]) __builtin_amdgcn_sched_barrier(0); }
            cons += ready_k ? 1u : 0u;
            look(nst, cons);
            const bool more = ballot64(ready) != 0ull;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; ++q) { A[q] = *reinterpret_cast<const lds_f4*>(src + 4 * q); RPW_CHAIN4(acc, B[q]) __builtin_amdgcn_sched_barrier(0); }
            asm volatile("" ::: "memory");                         // the slot's reads are done before its release is announced
            if (act_k) *(volatile lds_f32*)accw = acc;
            if (ready_k && row == 0) *(volatile lds_word*)consw = cons;
            if (!more) break;
        }
    }
}

// The server of ONE feeder (patch_fit_srv1_kernel: launches of few patches, where a patch's own time is what counts). Same
// hand-over words, same sums as srv_loop; what differs is what a slot costs. Measured on a lone wave (tools/microbench_srv1.hip,
// microbench_srv1b.hip): 64 dependent adds take 6.6 cycles each when the eight LDS reads of the next half slot are issued
// TOGETHER in front of 32 uninterrupted adds, 8.0 with one read between every four adds, 10.9 with a scalar branch in the
// middle of the slot, and a pair of exec-masked LDS stores costs ~125 cycles. So a slot here is: the published count (one
// LDS read), eight reads, 32 adds, eight reads of the NEXT slot's first half -- speculative: whether that slot had been
// published when they were issued is decided at the end from the count read in front of them -- 32 adds, two unmasked LDS
// stores (lanes without a result write into a dump block), one scalar branch.
struct Srv1Dump { uint32_t w[kWave]; float f[kWave]; };

__device__ __attribute__((noinline)) void srv1_loop(SrvShared* sv, Fit1Shared* sh, Srv1Dump* dump)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4u lds_u4;
    // (no raised priority here: the two waves of this workgroup sit on different SIMDs -- tools/microbench_placement.hip -- and
    // s_setprio 3 measured 1-2 % slower)
    const int lane = threadIdx.x & 63;
    // lane k < 6 adds row k. The lanes behind them (and the rows a pass does not use) add whatever their row of the ring holds:
    // nobody reads those accumulators, and it keeps every address of the slot a per-lane constant plus a scalar.
    const lds_f32* ringrow = (lds_f32*)sh->ring + (lane < 6 ? lane : 5) * kLdw;
    volatile lds_u4* fw = (volatile lds_u4*)sv->feed[0];
    volatile lds_word* pubw = (volatile lds_word*)&sv->feed[0][0];
    volatile lds_word* donep = (volatile lds_word*)&sv->done;
    volatile lds_f32* accw = lane < 6 ? (lds_f32*)&sv->acc[0][lane] : (lds_f32*)&dump->f[lane];
    volatile lds_word* consw = lane == 0 ? (lds_word*)&sv->cons[0] : (lds_word*)&dump->w[lane];
    uint32_t cons = 0;                                     // slots added so far (wave-uniform, like everything that decides below)
    float acc = 0.0f;
    v4f A[8], B[8];
    uint32_t start = 0, stride = 0, nsm = 0;
    const lds_f32* src = ringrow;
    bool have = false;                                     // the first half of slot `cons` is in A (requested while the slot before it was added)
    // The sixteen reads of a slot are written as inline assembly so that the next slot's first half can be read INTO the
    // registers the adds of this slot's first half have just released (a tied operand): left to the register allocator, the
    // loop-carried half went to fresh registers and was copied back by sixteen v_mov_b64 at the end of every slot (~130 cycles).
    // The compiler does not count these reads, so the waits for them are written out as well; its own waits (for the reads it
    // does see) can only wait longer, never shorter: LDS operations of a wave return in order.
#define RPW_LDS_READ8(DST, CONSTRAINT, ADDR, OFF)                                                                                         \
    asm volatile("ds_read_b128 %0, %8 offset:" #OFF "+0\n\tds_read_b128 %1, %8 offset:" #OFF "+16\n\tds_read_b128 %2, %8 offset:" #OFF "+32\n\t" \
                 "ds_read_b128 %3, %8 offset:" #OFF "+48\n\tds_read_b128 %4, %8 offset:" #OFF "+64\n\tds_read_b128 %5, %8 offset:" #OFF "+80\n\t" \
                 "ds_read_b128 %6, %8 offset:" #OFF "+96\n\tds_read_b128 %7, %8 offset:" #OFF "+112"                                         \
                 : CONSTRAINT(DST[0]), CONSTRAINT(DST[1]), CONSTRAINT(DST[2]), CONSTRAINT(DST[3]), CONSTRAINT(DST[4]), CONSTRAINT(DST[5]),    \
                   CONSTRAINT(DST[6]), CONSTRAINT(DST[7])                                                                                   \
                 : "v"((uint32_t)(uintptr_t)(ADDR)) : "memory")
#define RPW_LDS_WAIT8(N, X)                                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5]), "+v"(X[6]), "+v"(X[7]))
    for (;;) {                                             // one slot per iteration
        if (!have) {
            // ---- nothing usable prefetched: wait for a slot (or for the feeder to leave); a pass's layout is read here ----
            uint32_t layout;
            for (;;) {
                const v4u st = *fw;
                const uint32_t done = *donep;
                start = uni(st.y); layout = uni(st.z);
                if ((int32_t)(uni(st.x) - cons) > 0) break;
                if (uni(done) >= 1u) return;
                __builtin_amdgcn_s_sleep(0);
            }
            stride = layout & 0xffffu; nsm = layout >> 16;
            // (a pass begins here, never with a prefetched slot: the feeder starts the next pass only after it has seen the
            // last slot of this one added, and by then this wave has found nothing published and is polling)
            if (cons == start) acc = 0.0f;                         // Zero(), then +=
            src = ringrow + ((cons - start) & nsm) * stride;
            RPW_LDS_READ8(A, "=v", src, 0);
        }
        RPW_LDS_READ8(B, "=v", src, 128);                          // the slot's second half ...
        RPW_LDS_WAIT8(8, A);                                       // ... may still be on its way; the first half is there
#pragma unroll
        for (int q = 0; q < 8; ++q) RPW_CHAIN4(acc, A[q])
        __builtin_amdgcn_sched_barrier(0);
        ++cons;
        // The first half of the next slot, requested now whether or not that slot exists yet. The published count is read
        // in front of it (one wave's LDS operations are performed in order): if it covers the slot, the addends just
        // requested are the slot's -- the feeder writes a slot before it announces it. A count beyond `cons` can only belong
        // to THIS pass (see above), so stride and nsm still hold.
        const uint32_t pubv = *pubw;
        src = ringrow + ((cons - start) & nsm) * stride;
        asm volatile("" ::: "memory");
        RPW_LDS_READ8(A, "+v", src, 0);
        RPW_LDS_WAIT8(9, B);                                       // behind the second half: the count and the eight reads just issued
#pragma unroll
        for (int q = 0; q < 8; ++q) RPW_CHAIN4(acc, B[q])
        __builtin_amdgcn_sched_barrier(0);
        // every lane stores (its own address: a result word or a dump word), so no exec mask is set up
        *accw = acc;
        *consw = cons;
        have = (int32_t)(uni(pubv) - cons) > 0;                    // else: poll, then read the slot's first half again
    }
#undef RPW_LDS_READ8
#undef RPW_LDS_WAIT8
}

// The inliers' slot numbers, 64 at a time and in slot order, made from a mask plane (server forms: their fused pass writes
// no index list, and their sweeps read 0.125 B per slot of mask words instead of 2 B per inlier written and read again).
// Lane j keeps words wb + j and wb + 64 + j of the plane (8192 slots per block; a larger node reloads -- the one place where
// this waits for memory). Words are expanded a BATCH at a time by all 64 lanes: lane l takes piece l of the batch (PB bits:
// a byte of one of 8 words, or for nodes beyond 65536 slots a nibble of one of 4), fetched from the lane that holds the word
// (ds_bpermute); an exclusive scan of the pieces' popcounts (DPP) gives every lane the place of its first inlier, and its
// up to PB slot numbers go to consecutive entries of a circular LDS buffer, from where the sweep takes 64 consecutive
// entries per slot (one wave's LDS operations are performed in order). ~130 instructions per batch of 512 slots, i.e. ~30
// per 64 inliers at the bench's 56 %; word by word (one mbcnt per word) it was ~70, which a lone feeder could not hide.
// `count` must be the number of set bits of the node's words (the passes that write the masks return it); a mismatch cannot
// run away: the word cursor stops at the node's last word and the result is clamped to the node.
constexpr int kIdxBufBytes = 2048;                                 // 1024 16-bit or 512 32-bit entries
constexpr int kIdxBufAt = kSrvSweepSlots * kSlotFloats;            // behind the sweep ring's slots, in floats ...
static_assert(offsetof(Fit1Shared, hist) == sizeof(float) * kRingFloats, "... and on into hist (select_rank's, idle during a sweep)");
static_assert(kIdxBufAt * 4 + kIdxBufBytes <= (int)(sizeof(float) * kRingFloats + sizeof(uint32_t) * 256), "the index buffer must fit");
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v)
{
    // rows of 16 lanes: shifts by 1, 2, 4, 8 with zero fill; then lane 15 of a row to the row behind it, lane 31 to rows 2 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
template <int PB>                      // bits per piece: 8 (16-bit entries, nodes of up to 65536 slots) or 4 (32-bit entries)
struct IdxGen {
    static constexpr int PW = 64 / PB;                 // pieces per word
    static constexpr int WPB = kWave / PW;             // words per batch
    static constexpr uint32_t kEntries = PB == 8 ? (uint32_t)kIdxBufBytes / 2u : (uint32_t)kIdxBufBytes / 4u;
    static_assert(kEntries >= (uint32_t)(kWave * PB + kWave), "a batch's inliers next to the unread rest of a slot");
    g_u64* words;
    uint32_t ib;                       // LDS byte address of the buffer
    uint32_t nw, count, n;
    uint32_t wc, wb, filled;           // next word to expand, first word of the block in W0 / W1, inliers expanded so far (wave-uniform)
    uint32_t W0lo, W0hi, W1lo, W1hi;
    int lane;
    // (the empty asm "uses" the words right here: the wait for the two loads is placed in this rarely executed block, where
    // it costs one memory latency per 8192 slots; without it the wait sat in front of EVERY expansion as vmcnt(0), i.e.
    // every call drained the sweep's prefetched point loads)
    __device__ __forceinline__ void load_block(uint32_t w0)
    {
        wb = w0;
        const uint32_t a = w0 + (uint32_t)lane, b = a + 64u;
        const unsigned long long x = words[a < nw ? a : nw - 1u], y = words[b < nw ? b : nw - 1u];
        W0lo = (uint32_t)x; W0hi = (uint32_t)(x >> 32); W1lo = (uint32_t)y; W1hi = (uint32_t)(y >> 32);
        asm volatile("" : "+v"(W0lo), "+v"(W0hi), "+v"(W1lo), "+v"(W1hi));
    }
    __device__ __forceinline__ void begin(g_u64* plane, uint32_t n_, uint32_t count_, lds_f32* ring, int ln)
    {
        words = plane; n = n_; count = count_; nw = (n_ + 63u) >> 6; lane = ln;
        ib = (uint32_t)(uintptr_t)ring + (uint32_t)kIdxBufAt * 4u;
        wc = 0; filled = 0;
        load_block(0);
    }
    __device__ __forceinline__ void expand()
    {
        if (wc - wb >= 128u) load_block(wc);
        const uint32_t k = wc - wb;                                  // a multiple of WPB: a batch lies in W0 or in W1
        const uint32_t w = (uint32_t)lane / (uint32_t)PW, piece = (uint32_t)lane % (uint32_t)PW;
        const int src = (int)((((k & 63u) + w) & 63u) * 4u);
        const bool upper = k >= 64u;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(upper ? W1lo : W0lo));
        const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(upper ? W1hi : W0hi));
        const uint32_t half = piece < (uint32_t)(PW / 2) ? lo : hi;
        uint32_t bits = (half >> ((piece % (uint32_t)(PW / 2)) * (uint32_t)PB)) & ((1u << PB) - 1u);
        if (wc + w >= nw) bits = 0u;                                 // (the clamped loads repeat the node's last word)
        const uint32_t cnt = (uint32_t)__builtin_popcount(bits);
        const uint32_t incl = wave_incl_scan_dpp(cnt);
        uint32_t pos = filled + incl - cnt;
        const uint32_t slot0 = wc * (uint32_t)kWave + (uint32_t)lane * (uint32_t)PB;      // the slot of the piece's bit 0
#pragma unroll
        for (int it = 0; it < PB; ++it) {
            if (ballot64(bits != 0u) == 0ull) break;
            if (bits) {
                const uint32_t v = slot0 + (uint32_t)(__builtin_ctz(bits));
                const uint32_t e = pos & (kEntries - 1u);
                if (PB == 8) *(__attribute__((address_space(3))) uint16_t*)(uintptr_t)(ib + e * 2u) = (uint16_t)v;
                else         *(lds_word*)(uintptr_t)(ib + e * 4u) = v;
                ++pos;
                bits &= bits - 1u;
            }
        }
        filled += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        wc += (uint32_t)WPB;
    }
    // the slot numbers of inliers [64 u, 64 u + 64) (lanes past the last inlier repeat it); u must not decrease between calls
    __device__ __forceinline__ uint32_t next(uint32_t u)
    {
        const uint32_t end = (u + 1u) * (uint32_t)kWave, target = end < count ? end : count;
        while (filled < target && wc < nw) expand();
        asm volatile("" ::: "memory");
        const uint32_t j = u * (uint32_t)kWave + (uint32_t)lane, last = count - 1u, jc = j < last ? j : last;
        const uint32_t e = jc & (kEntries - 1u);
        uint32_t idx;
        if (PB == 8) idx = (uint32_t)*(__attribute__((address_space(3))) const uint16_t*)(uintptr_t)(ib + e * 2u);
        else         idx = *(const lds_word*)(uintptr_t)(ib + e * 4u);
        asm volatile("" ::: "memory");
        return idx < n ? idx : n - 1u;
    }
};

// feeder half of a chain sweep: slot s goes to ring slot s % kSrvSweepSlots once the
// server has added slot s - kSrvSweepSlots
template <int MODE, int SRC>        // SRC: 0 = every slot of the node; 1 / 3 = the inliers of plane nd.mcur (node of up to / beyond 65536 slots); 2 / 4 = the index list (bytes per entry)
__device__ __forceinline__ Sums6 srv_sweep_loop(const NodeRef& nd, uint32_t count, const SweepPrm& prm, lds_f32* ring, const SrvPort& pt, int lane)
{
    constexpr int K = SweepRows<MODE>::K;
    typedef float f2 __attribute__((ext_vector_type(2)));
    const uint32_t nslots = (count + kWave - 1) / kWave;
    const uint32_t last = count - 1u;
    typedef __attribute__((address_space(1))) const char g_cchar;
    g_cchar* il = reinterpret_cast<g_cchar*>(nd.il);
    constexpr bool MASKED = SRC == 1 || SRC == 3;
    IdxGen<(SRC == 3 ? 4 : 8)> gen;
    if (MASKED) gen.begin(nd.mcur, nd.n, count, ring, lane);
    auto index_of = [&](uint32_t u) -> uint32_t {
        if (MASKED) return gen.next(u);
        const uint32_t j = u * kWave + lane, jc = j < last ? j : last;
        return SRC ? il_load<(SRC == 2 ? 2 : 4)>(il, jc) : jc;
    };
    auto point_of = [&](uint32_t idx) -> X3 {
        X3 q; q.x = ldg_f32(nd.x, idx); q.y = ldg_f32(nd.y, idx);
        q.z = (MODE == kSweepXYZ || MODE == kSweepCov || MODE == kSweepDist || MODE == kSweepRange) ? ldg_f32(nd.z, idx) : 0.0f;
        return q;
    };
    const uint32_t start = srv_begin(pt, srv_layout((uint32_t)kSlotFloats, (uint32_t)kSrvSweepSlots), (uint32_t)K, lane);
    uint32_t seen = start;
    // kSrvSweepDepth STATIC register sets, used round robin and reloaded right after use (the loop is unrolled that many times):
    // P[k] holds the points of the next slot == k (mod depth), I[k] the indices of the slot `depth` behind it -- points `depth`
    // slots ahead of their addends, indices `depth` slots ahead of the point loads. Nothing is ever copied from one set to
    // another: a copy reads its source, i.e. waits for a load issued one slot earlier, and the whole pipeline is then one slot
    // deep (that was the case here until round 3; at full occupancy the covariance sweep ran 20 % slower for it).
    constexpr int D = kSrvSweepDepth;
    X3 P[D]; uint32_t I[D];
    float zmn = 3.402823466e+38f, zmx = -3.402823466e+38f;      // kSweepRange also returns the z range of the elements (v[1], v[2])
#pragma unroll
    for (int k = 0; k < D; ++k) I[k] = index_of((uint32_t)k);
#pragma unroll
    for (int k = 0; k < D; ++k) P[k] = point_of(I[k]);
#pragma unroll
    for (int k = 0; k < D; ++k) I[k] = index_of((uint32_t)(D + k));
    auto slot = [&](uint32_t s, X3& P, uint32_t& I, auto full) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full)::value;              // all 64 columns of the slot are elements (no padding to write)
        float v[6];
        const X3 q = P;
        if (MODE == kSweepCov) {
            const f2 dxy = f2{q.x, q.y} - f2{prm.a, prm.b};
            const float dz = q.z - prm.c;
            const f2 a = dxy * f2{dxy.x, dxy.x};            // dx*dx, dy*dx
            const f2 b = f2{dxy.y, dz} * f2{dxy.y, dxy.x};  // dy*dy, dz*dx
            const f2 c = f2{dz, dz} * f2{dxy.y, dz};        // dz*dy, dz*dz
            v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
        } else if (MODE == kSweepXYZ) { v[0] = q.x; v[1] = q.y; v[2] = q.z; }
        else if (MODE == kSweepDist)  { v[0] = plane_dist(q.x, q.y, q.z, prm.a, prm.b, prm.c, prm.d, prm.e, prm.f); }
        else if (MODE == kSweepRange) { v[0] = sqrt_rn(q.x * q.x + q.y * q.y); zmn = fminf(zmn, q.z); zmx = fmaxf(zmx, q.z); }      // (+ the node's z range: order-free)
        else if (MODE == kSweepXY)    { v[0] = q.x; v[1] = q.y; }
        else { const float dx = q.x - prm.a, dy = q.y - prm.b; v[0] = dx * dx; v[1] = dy * dy; }
        if (s >= (uint32_t)kSrvSweepSlots) srv_wait(pt, start + s - (uint32_t)(kSrvSweepSlots - 1), seen);
        lds_f32* dst = ring + (s % (uint32_t)kSrvSweepSlots) * kSlotFloats + lane;
        const bool live = FULL || s * kWave + lane < count;   // columns past the end: -0.0f, the exact additive identity
#pragma unroll
        for (int k = 0; k < K; ++k) dst[k * kLdw] = live ? v[k] : -0.0f;
        srv_publish(pt, start + s + 1u, lane);
        P = point_of(I);                                      // slot s + D
        I = index_of(s + 2u * (uint32_t)D);
    };
    uint32_t s = 0;
    const uint32_t nfull = count / (uint32_t)kWave;               // slots without padding
    for (; s + (uint32_t)D <= nfull; s += (uint32_t)D) {
#pragma unroll
        for (int k = 0; k < D; ++k) slot(s + (uint32_t)k, P[k], I[k], yes());
    }
#pragma unroll
    for (int k = 0; k < D; ++k) if (s + (uint32_t)k < nslots) slot(s + (uint32_t)k, P[k], I[k], no());
    srv_wait(pt, start + nslots, seen);
    Sums6 out;
#pragma unroll
    for (int k = 0; k < 6; ++k) out.v[k] = k < K ? *(volatile lds_f32*)(pt.accp + k) : 0.0f;
    if (MODE == kSweepRange) { out.v[1] = wave_min(zmn); out.v[2] = wave_max(zmx); }
    return out;
}

// The server forms' passes are real functions (not inlined), and what does not fit the register convention travels over
// the STACK: a 24-byte parameter struct costs six dwords per LANE, stored by the caller and loaded by the callee -- 3 KB of
// scratch traffic per call, two calls per refit, ~6 GB per bench launch that L2 cannot hold between the two ends (its 4 MB
// turn over in microseconds). So the caller leaves the arguments in its LDS block and passes the block's address.
// Word layout of the block: 0-13 the node's seven pointers (x, y, z, mnew, mcur, mprev, il), 14 n, 15 count, 16 mode, 17-23 p[0..6].
// (Packed and unpacked word by word: a struct copied through a cast has its address taken and lives on the stack itself.)
struct PassArgs { NodeRef nd; uint32_t count; int32_t mode; float p[7]; };
__device__ __forceinline__ uint32_t pass_args_of(lds_f32* ring)       // the ring is the first member of the owner's Fit1Shared
{
    return (uint32_t)(uintptr_t)ring + (uint32_t)offsetof(Fit1Shared, args);
}
__device__ __forceinline__ void pass_args_put(lds_f32* ring, const NodeRef& nd, uint32_t count, int mode, const float* p, int np)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4u lds_u4;
    auto lo = [](const void __attribute__((address_space(1)))* q) { return (uint32_t)(unsigned long long)q; };
    auto hi = [](const void __attribute__((address_space(1)))* q) { return (uint32_t)((unsigned long long)q >> 32); };
    auto fw = [&](int k) { return k < np ? __float_as_uint(p[k]) : 0u; };
    if ((threadIdx.x & 63) == 0) {
        lds_u4* dst = (lds_u4*)(uintptr_t)pass_args_of(ring);
        dst[0] = v4u{lo(nd.x), hi(nd.x), lo(nd.y), hi(nd.y)};
        dst[1] = v4u{lo(nd.z), hi(nd.z), lo(nd.mnew), hi(nd.mnew)};
        dst[2] = v4u{lo(nd.mcur), hi(nd.mcur), lo(nd.mprev), hi(nd.mprev)};
        dst[3] = v4u{lo(nd.il), hi(nd.il), nd.n, count};
        dst[4] = v4u{(uint32_t)mode, fw(0), fw(1), fw(2)};
        dst[5] = v4u{fw(3), fw(4), fw(5), fw(6)};
    }
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ PassArgs pass_args_get(lds_f32* ring)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4u lds_u4;
    const lds_u4* src = (const lds_u4*)(uintptr_t)pass_args_of(ring);
    const v4u w0 = src[0], w1 = src[1], w2 = src[2], w3 = src[3], w4 = src[4], w5 = src[5];
    auto ptr = [](uint32_t l, uint32_t h) { return ((unsigned long long)uni(h) << 32) | (unsigned long long)uni(l); };
    PassArgs a;
    a.nd.x = (g_cf32*)ptr(w0.x, w0.y); a.nd.y = (g_cf32*)ptr(w0.z, w0.w); a.nd.z = (g_cf32*)ptr(w1.x, w1.y);
    a.nd.mnew = (g_u64*)ptr(w1.z, w1.w); a.nd.mcur = (g_u64*)ptr(w2.x, w2.y); a.nd.mprev = (g_u64*)ptr(w2.z, w2.w);
    a.nd.il = (g_u32*)ptr(w3.x, w3.y); a.nd.n = uni(w3.z); a.count = uni(w3.w);
    a.mode = (int32_t)uni(w4.x);
    a.p[0] = uni(__uint_as_float(w4.y)); a.p[1] = uni(__uint_as_float(w4.z)); a.p[2] = uni(__uint_as_float(w4.w));
    a.p[3] = uni(__uint_as_float(w5.x)); a.p[4] = uni(__uint_as_float(w5.y)); a.p[5] = uni(__uint_as_float(w5.z)); a.p[6] = uni(__uint_as_float(w5.w));
    return a;
}

// MASKED (the batch form): the inliers come from the mask plane; else (a server per patch) from the index list the fused pass
// wrote. Turning mask words into slot numbers costs the feeder ~70 instructions per slot of 64 inliers: among seven feeders
// and two workgroups per CU they disappear and the bench batch gains 2.4 % from the 2 B per inlier no longer written and
// read; a LONE feeder issues one instruction per ~6 cycles, its covariance sweep went from 9.7 to 16.7 cycles per inlier
// (the server waits for it) and a single frame from 2.48 to 3.13 ms -- so the latency form keeps the list.
template <bool MASKED>
__device__ __attribute__((noinline)) Sums6 srv_chain_sweep_call(lds_f32* ring, SrvShared* sv, int f)
{
    const int lane = threadIdx.x & 63;
    ring = (lds_f32*)(uintptr_t)uni((uint32_t)(uintptr_t)ring);
    const PassArgs a = pass_args_get(ring);
    const NodeRef nd = a.nd;
    const uint32_t count = a.count;
    const int mode = a.mode;
    SweepPrm prm; prm.a = a.p[0]; prm.b = a.p[1]; prm.c = a.p[2]; prm.d = a.p[3]; prm.e = a.p[4]; prm.f = a.p[5];
    sv = uni_ptr(sv);
    const SrvPort pt = srv_port(sv, (int)uni((uint32_t)f), ring);
    Sums6 out;
#pragma unroll
    for (int k = 0; k < 6; ++k) out.v[k] = 0.0f;
    if (count == 0) return out;
    const bool narrow = nd.n <= 65536u;                // the width the pass that wrote the list used
    const bool listed = MASKED && kListMaxSlots != 0u && nd.n <= kListMaxSlots;      // a small node of the batch form: 16-bit list
#define RPW_SRV_INDEXED(M) (listed ? srv_sweep_loop<M, 2>(nd, count, prm, ring, pt, lane)                   \
                          : (narrow ? srv_sweep_loop<M, (MASKED ? 1 : 2)>(nd, count, prm, ring, pt, lane)   \
                                    : srv_sweep_loop<M, (MASKED ? 3 : 4)>(nd, count, prm, ring, pt, lane)))
    switch (mode) {
    case kSweepCov:   return RPW_SRV_INDEXED(kSweepCov);
    case kSweepXYZ:   return RPW_SRV_INDEXED(kSweepXYZ);
    case kSweepDist:  return RPW_SRV_INDEXED(kSweepDist);
    case kSweepRange: return srv_sweep_loop<kSweepRange, 0>(nd, count, prm, ring, pt, lane);
    case kSweepXY:    return srv_sweep_loop<kSweepXY, 0>(nd, count, prm, ring, pt, lane);
    default:          return srv_sweep_loop<kSweepVarXY, 0>(nd, count, prm, ring, pt, lane);
    }
#undef RPW_SRV_INDEXED
}
template <bool MASKED>
__device__ __forceinline__ Sums6 srv_chain_sweep(const NodeRef& nd, uint32_t count, int mode, const SweepPrm& prm, lds_f32* ring, SrvShared* sv, int f)
{
    const float p[6] = {prm.a, prm.b, prm.c, prm.d, prm.e, prm.f};
    pass_args_put(ring, nd, count, mode, p, 6);
    return srv_chain_sweep_call<MASKED>(ring, sv, f);
}

// the fused pass's chain, handed to the server: "consume" publishes, "finish" pads, publishes and collects
struct SrvChain {
    SrvPort pt;
    uint32_t start, seen;
    int lane;
    float sums[3];
    __device__ __forceinline__ void begin(const SrvPort& port, int ln)
    {
        pt = port; lane = ln;
        start = srv_begin(pt, srv_layout((uint32_t)kFSlotFloats, (uint32_t)kFSlots), 3u, lane);
        seen = start;
    }
    // the coming chunks write addends below position `upto`: their ring slots must have been added
    __device__ __forceinline__ void reserve(uint32_t upto)
    {
        const uint32_t smax = (upto - 1u) >> 6;
        if (smax >= (uint32_t)kFSlots) srv_wait(pt, start + smax - (uint32_t)(kFSlots - 1), seen);
    }
    __device__ __forceinline__ void consume(const lds_f32*, uint32_t avail) { srv_publish(pt, start + avail, lane); }
    __device__ __forceinline__ void finish(lds_f32* ring, const lds_f32*, uint32_t cursor, int)
    {
        sums[0] = sums[1] = sums[2] = 0.0f;
        if (cursor == 0) return;
        if ((cursor & 63u) && (uint32_t)lane >= (cursor & 63u)) {
            lds_f32* w = ring + fused_ring_pos((cursor & ~63u) + (uint32_t)lane);
            w[0] = -0.0f; w[kLdw] = -0.0f; w[2 * kLdw] = -0.0f;
        }
        const uint32_t total = (cursor + 63u) >> 6;
        srv_publish(pt, start + total, lane);
        srv_wait(pt, start + total, seen);
#pragma unroll
        for (int k = 0; k < 3; ++k) sums[k] = *(volatile lds_f32*)(pt.accp + k);
    }
    __device__ __forceinline__ float result(int k) const { return sums[k]; }
};

template <bool LIST>               // LIST: the pass also writes the index list (the latency form's sweeps read it)
__device__ __attribute__((noinline)) FusedOut srv_fused_pass_call(lds_f32* ring, SrvShared* sv, int f)
{
    const int lane = threadIdx.x & 63;
    ring = (lds_f32*)(uintptr_t)uni((uint32_t)(uintptr_t)ring);
    const PassArgs a = pass_args_get(ring);
    const NodeRef nd = a.nd;
    const int mode = a.mode;
    MaskPrm prm; prm.cx = a.p[0]; prm.cy = a.p[1]; prm.cz = a.p[2]; prm.nx = a.p[3]; prm.ny = a.p[4]; prm.nz = a.p[5]; prm.thr = a.p[6];
    sv = uni_ptr(sv);
    SrvChain ch;
    ch.begin(srv_port(sv, (int)uni((uint32_t)f), ring), lane);
    const bool narrow = nd.n <= 65536u;
    const bool listed = !LIST && kListMaxSlots != 0u && nd.n <= kListMaxSlots;      // a small node of the batch form keeps a (16-bit) list
    if (mode == kMaskPlane)
        return !LIST ? (listed ? w1_fused_loop<kMaskPlane, 2>(nd, prm, ring, lane, ch) : w1_fused_loop<kMaskPlane, 0>(nd, prm, ring, lane, ch))
                     : (narrow ? w1_fused_loop<kMaskPlane, (LIST ? 2 : 0)>(nd, prm, ring, lane, ch) : w1_fused_loop<kMaskPlane, (LIST ? 4 : 0)>(nd, prm, ring, lane, ch));
    return !LIST ? (listed ? w1_fused_loop<kMaskSeed, 2>(nd, prm, ring, lane, ch) : w1_fused_loop<kMaskSeed, 0>(nd, prm, ring, lane, ch))
                 : (narrow ? w1_fused_loop<kMaskSeed, (LIST ? 2 : 0)>(nd, prm, ring, lane, ch) : w1_fused_loop<kMaskSeed, (LIST ? 4 : 0)>(nd, prm, ring, lane, ch));
}
template <bool LIST>
__device__ __forceinline__ FusedOut srv_fused_pass(const NodeRef& nd, int mode, const MaskPrm& prm, lds_f32* ring, SrvShared* sv, int f)
{
    const float p[7] = {prm.cx, prm.cy, prm.cz, prm.nx, prm.ny, prm.nz, prm.thr};
    pass_args_put(ring, nd, 0u, mode, p, 7);
    return srv_fused_pass_call<LIST>(ring, sv, f);
}

// workgroup set-up and the feeder's good-bye
constexpr size_t kSrvLdsBytes = sizeof(SrvShared) + (size_t)kSrvFeeders * sizeof(Fit1Shared);
static_assert(sizeof(SrvShared) % 16 == 0, "the feeders' blocks follow the hand-over block, 16-byte aligned");
__device__ __forceinline__ void srv_init(SrvShared* sv)
{
    if (threadIdx.x < 64) {
        const int t = threadIdx.x;
        if (t < 32) (&sv->feed[0][0])[t] = 0u;
        (&sv->acc[0][0])[t] = 0.0f;
        sv->zero[t] = -0.0f;
        if (t < 8) sv->cons[t] = 0u;
        if (t == 0) { sv->done = 0u; sv->error = 0u; }
    }
    __syncthreads();
}
__device__ __forceinline__ void srv_leave(SrvShared* sv, int lane)
{
    wave_sync();
    if (lane == 0) atomicAdd(&sv->done, 1u);
}

// work item -> patch, largest size bucket first; false when the launch has fewer patches than that
__device__ __forceinline__ bool patch_of_item(const DevParams& p, uint32_t item, int lane, uint32_t& linear)
{
    const uint32_t cnt = p.bucket_count[kBuckets - 1 - lane];
    const uint32_t incl = wave_incl_scan(cnt);
    const uint32_t total = __shfl(incl, 63);
    if (item >= total) return false;
    const unsigned long long bm = ballot64(item < incl);
    const int l = __ffsll((long long)bm) - 1;
    const uint32_t before = __shfl(incl - cnt, l);
    linear = p.bucket_list[(size_t)(kBuckets - 1 - l) * p.list_stride + (item - before)];
    return true;
}

// One patch, on the wave that owns it. W = 1: alone; W = 3: as feeder `feeder` of a chain server of seven feeders (the sweeps
// take the inliers from the mask planes); W = 4: as the feeder of a server of its own (index list, like W = 1).
template <int W>
__device__ __forceinline__ void fit_patch(const DevParams& p, Fit1Shared& sh, uint32_t linear, int lane, SrvShared* sv, int feeder)
{
#define W1_PH(IDX, CNT, ...) { __VA_ARGS__; }
    // the passes that run a chain: everything on this wave (single-wave form), or the feeding half here and the adds on a
    // server wave
#define W_SWEEP(ND, M, MODE, INDEXED, PRM) (W == 3 ? srv_chain_sweep<true>(ND, M, MODE, PRM, ring, sv, feeder) : (W == 4 ? srv_chain_sweep<false>(ND, M, MODE, PRM, ring, sv, feeder) : w1_chain_sweep(ND, M, MODE, INDEXED, PRM, ring)))
#define W_FUSED(ND, MODE, PRM) (W == 3 ? srv_fused_pass<false>(ND, MODE, PRM, ring, sv, feeder) : (W == 4 ? srv_fused_pass<true>(ND, MODE, PRM, ring, sv, feeder) : w1_fused_pass(ND, MODE, PRM, ring)))
#define W_LEAVE() {}
    // Everything the node loop carries across its pass calls is wave-uniform; said so (v_readfirstlane), it lives in scalar
    // registers, whose spills go to lanes of a vector register -- a uniform value left in a VECTOR register across a call is
    // spilled to scratch memory, 256 bytes per dword and wave, and read back from there call after call.
    linear = uni(linear);
    const int frame = (int)(linear / (uint32_t)p.P), patch = (int)(linear % (uint32_t)p.P);
    PatchDesc pd = p.patches[linear];
    pd.begin = uni(pd.begin); pd.n = uni(pd.n);
    const uint32_t fo = uni(p.frame_off[frame]);

    W1 cx; cx.sx = p.sx; cx.sy = p.sy; cx.sz = p.sz; cx.il = reinterpret_cast<uint32_t*>(p.ax); cx.sh = &sh;
    cx.base = pd.begin; cx.lane = lane; cx.addends = 0ull;
    const rpw_config& cfg = p.cfg;
    uint32_t* stk = sh.stk;

    // zone < 3: the reference returns {{}, cleaned} before looking at any patch (recursive_patchwork.cpp:339-341)
    if (p.stats[frame].n_zone < 3u) {
        for (uint32_t i = lane; i < pd.n; i += kFit1Threads) p.labels[fo + p.sidx[pd.begin + i]] = RPW_LABEL_NONGROUND;
        W_LEAVE()
        return;
    }

    lds_f32* ring = (lds_f32*)sh.ring;            // address-space cast: the passes take the ring as an LDS pointer (ds_ instructions, not flat)
    const SweepPrm no_prm = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    // ---- mean distance of the patch (recursive_patchwork.cpp:383-387): sequential sum of d ------
    float mean_dist, root_zmn, root_zmx;
    {
        NodeRef root; root.x = (g_cf32*)(p.sx + cx.base); root.y = (g_cf32*)(p.sy + cx.base); root.z = (g_cf32*)(p.sz + cx.base);
        root.mnew = root.mcur = root.mprev = nullptr; root.il = (g_u32*)(cx.il + cx.base); root.n = pd.n;      // (the range sweep touches no mask)
        Sums6 sum;
        W1_PH(1, pd.n, sum = W_SWEEP(root, pd.n, kSweepRange, 0, no_prm))
        cx.addends += pd.n;
        mean_dist = uni(sum.v[0]) / (float)pd.n;
        root_zmn = uni(sum.v[1]); root_zmx = uni(sum.v[2]);      // the patch's z range rides on the same sweep (what the root node's flat test needs)
    }
    const float rel_dist = uni(mean_dist / cfg.filtering_radius);
    const float threshold = uni(cfg.th_dist * (1.0f + 0.2f * rel_dist));

    PatchAcc pa; pa.nodes = 0; pa.splits = 0; pa.maxdepth = 0; pa.ground = 0; pa.iters = 0;
    bool permuted = false;
    // the patch's words of the three mask planes (patch p of the batch starts at word (begin >> 6) + p: ceil(n / 64) words fit in front of
    // the next patch's); mC = the current mask, mQ = the one before it, mF = free for the next pass to write
    g_u64* mC = (g_u64*)p.mbits + ((size_t)(pd.begin >> 6) + linear);
    g_u64* mQ = mC + p.mplane;
    g_u64* mF = mQ + p.mplane;

    int sp = 0;
    if (lane == 0) { stk[0] = 0; stk[1] = pd.n; stk[2] = 0; }
    sp = 1;
    wave_sync();

    while (sp > 0) {
        --sp;
        const uint32_t s = uni(stk[3 * sp]), n = uni(stk[3 * sp + 1]);
        const int depth = (int)uni(stk[3 * sp + 2]);
        wave_sync();
        pa.nodes++;
        if ((uint32_t)depth > pa.maxdepth) pa.maxdepth = (uint32_t)depth;
        const float* gx = p.sx + cx.base + s;
        const float* gy = p.sy + cx.base + s;
        const float* gz = p.sz + cx.base + s;
        const uint32_t* gi = p.sidx + cx.base + s;
        uint32_t* gl = cx.il + cx.base + s;

        // ---- early outs (recursive_patchwork.cpp:111-140) ---------------------------------------
        int leaf_kind = -1;
        if (n < 3 || depth > cfg.max_split_depth) leaf_kind = RPW_NODE_REJECT;
        if (leaf_kind < 0) {
            float xmn = 3.402823466e+38f, xmx = -3.402823466e+38f, ymn = xmn, ymx = xmx, zmn = xmn, zmx = xmx;
            if (depth == 0) {                         // the root: its area is never tested (recursive_patchwork.cpp:127), its z range is known
                xmn = xmx = ymn = ymx = 0.0f; zmn = root_zmn; zmx = root_zmx;
            } else {
                for (uint32_t i0 = lane; i0 < n; i0 += 4 * kFit1Threads) {
                    float x[4], y[4], z[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const uint32_t i = i0 + u * kFit1Threads, ic = i < n ? i : n - 1u; x[u] = gx[ic]; y[u] = gy[ic]; z[u] = gz[ic]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        xmn = fminf(xmn, x[u]); xmx = fmaxf(xmx, x[u]); ymn = fminf(ymn, y[u]); ymx = fmaxf(ymx, y[u]);
                        zmn = fminf(zmn, z[u]); zmx = fmaxf(zmx, z[u]);
                    }
                }
                xmn = wave_min(xmn); xmx = wave_max(xmx); ymn = wave_min(ymn); ymx = wave_max(ymx);
                zmn = wave_min(zmn); zmx = wave_max(zmx);
            }
            const float area = (xmx - xmn) * (ymx - ymn);
            if (area < 25.0f && depth > 0) leaf_kind = RPW_NODE_SMALL;
            else if ((zmx - zmn) < 0.05f && n > 10) leaf_kind = RPW_NODE_FLAT;
        }
        if (leaf_kind >= 0) {
            const uint8_t lab = leaf_kind == RPW_NODE_REJECT ? RPW_LABEL_NONGROUND : RPW_LABEL_GROUND;
            for (uint32_t i = lane; i < n; i += kFit1Threads) p.labels[fo + gi[i]] = lab;
            if (leaf_kind != RPW_NODE_REJECT) pa.ground += n;
            emit_node(p, lane, frame, patch, depth, s, n, leaf_kind, 0, leaf_kind == RPW_NODE_REJECT ? 0u : n, -1, 0.0f, nullptr);
            continue;
        }

        // ---- seeds (recursive_patchwork.cpp:149-182) ---------------------------------------------
        float z_th;
        if (cfg.adaptive_seed_height) {
            z_th = cfg.sensor_height + 0.2f * rel_dist;
        } else {
            const uint32_t idx = (uint32_t)(0.1f * (float)n);     // static_cast<size_t>(0.1f * size)
            z_th = w1_select_rank(cx, s, n, 2, idx < n ? idx : n - 1) + cfg.th_seeds;
        }
        NodeRef nd; nd.x = (g_cf32*)gx; nd.y = (g_cf32*)gy; nd.z = (g_cf32*)gz; nd.mnew = mF; nd.mcur = mC; nd.mprev = mQ; nd.il = (g_u32*)gl; nd.n = n;
        auto rotate = [&]() { g_u64* t = mQ; mQ = mC; mC = mF; mF = t; nd.mnew = mF; nd.mcur = mC; nd.mprev = mQ; };      // a pass wrote mF: it is the current mask now
        MaskPrm mp; mp.cx = mp.cy = mp.cz = 0.0f; mp.nx = mp.ny = 0.0f; mp.nz = 1.0f; mp.thr = z_th;
        float cs[3];                                   // centroid sums of the current mask
        uint32_t count;
        {   // seed mask + index list + centroid sums of the seeds, one pass
            FusedOut fo_;
            W1_PH(2, n, fo_ = W_FUSED(nd, kMaskSeed, mp))
            count = uni(fo_.count); cs[0] = uni(fo_.sx); cs[1] = uni(fo_.sy); cs[2] = uni(fo_.sz);
            cx.addends += count;
            rotate();
        }
        if (count < 3) {
            const Sel3 sel = w1_three_lowest(cx, s, n);
            for (uint32_t i = lane; i < n + (uint32_t)lane; i += kFit1Threads)          // (whole chunks: every lane takes part in the ballot)
                mC[i >> 6] = ballot64(i < n && (i == sel.v[0] || i == sel.v[1] || i == sel.v[2]));
            // the index list in slot order: the three slots, ascending (entries as wide as the passes read them)
            uint32_t a = sel.v[0], b = sel.v[1], c = sel.v[2], t;
            if (a > b) { t = a; a = b; b = t; }
            if (b > c) { t = b; b = c; c = t; }
            if (a > b) { t = a; a = b; b = t; }
            if (lane == 0) {
                if (n <= 65536u) { uint16_t* g16 = reinterpret_cast<uint16_t*>(gl); g16[0] = (uint16_t)a; g16[1] = (uint16_t)b; g16[2] = (uint16_t)c; }
                else { gl[0] = a; gl[1] = b; gl[2] = c; }
            }
            count = 3;
            wave_sync();                           // index list and mask bytes written by this wave: drain before the reads
            Sums6 r;
            W1_PH(3, 3, r = W_SWEEP(nd, 3, kSweepXYZ, 1, no_prm))
            cx.addends += 3;
            cs[0] = uni(r.v[0]); cs[1] = uni(r.v[1]); cs[2] = uni(r.v[2]);
        }
        wave_sync();

        // ---- iterative refit (recursive_patchwork.cpp:186-217) ------------------------------------
        PlaneD pl; pl.cx = pl.cy = pl.cz = 0.0f; pl.nx = 0.0f; pl.ny = 0.0f; pl.nz = 1.0f; pl.residual = 3.402823466e+38f;
        auto plane_of = [&](uint32_t m) {               // covariance sweep over the index list + eigenvector
            const float fm = (float)m;
            SweepPrm c; c.a = uni(cs[0] / fm); c.b = uni(cs[1] / fm); c.c = uni(cs[2] / fm); c.d = c.e = c.f = 0.0f;
            Sums6 cv;
            W1_PH(4, m, cv = W_SWEEP(nd, m, kSweepCov, 1, c))
            cx.addends += m;
            const float fm1 = (float)(m - 1);
            pl.cx = c.a; pl.cy = c.b; pl.cz = c.c;
            N3 nrm;
            W1_PH(5, 1, nrm = w1_normal(cv.v[0] / fm1, cv.v[1] / fm1, cv.v[2] / fm1, cv.v[3] / fm1, cv.v[4] / fm1, cv.v[5] / fm1))
            pl.nx = uni(nrm.x); pl.ny = uni(nrm.y); pl.nz = uni(nrm.z);
        };
        auto residual_of = [&](uint32_t m, bool older = false) -> float {   // sum of |d| over the inliers / m (recursive_patchwork.cpp:98-104)
            SweepPrm q; q.a = pl.cx; q.b = pl.cy; q.c = pl.cz; q.d = pl.nx; q.e = pl.ny; q.f = pl.nz;
            Sums6 r;
            NodeRef ndr = nd;
            if (older) ndr.mcur = nd.mprev;             // (server forms: the sweep's elements are the inliers of plane mcur)
            W1_PH(6, m, r = W_SWEEP(ndr, m, kSweepDist, 1, q))
            cx.addends += m;
            return uni(uni(r.v[0]) / (float)m);
        };
        int iters = 0;
        int final_bit = 0;
        bool have_plane = false, have_residual = false;   // pl / pl.residual belong to exactly the final mask
        bool list_is_final = true;                        // the index list holds the final mask's inliers
        for (int iter = 0; iter < cfg.max_iter; ++iter) {
            if (count < 3) break;
            plane_of(count);
            ++iters;
            mp.cx = pl.cx; mp.cy = pl.cy; mp.cz = pl.cz; mp.nx = pl.nx; mp.ny = pl.ny; mp.nz = pl.nz; mp.thr = threshold;
            FusedOut mo;                                         // new mask + its index list + its centroid sums
            W1_PH(2, n, mo = W_FUSED(nd, kMaskPlane, mp))
            rotate();
            mo.count = uni(mo.count); mo.changed = uni(mo.changed); mo.differs_prev = uni(mo.differs_prev);
            mo.sx = uni(mo.sx); mo.sy = uni(mo.sy); mo.sz = uni(mo.sz);
            const uint32_t newcount = mo.count;
            cx.addends += newcount;
            const bool any_changed = mo.changed != 0, any_differs_prev = mo.differs_prev != 0;
            wave_sync();
            if (!any_changed) {                                  // new_mask == ground_mask: converged
                pl.residual = residual_of(count);
                have_plane = true; have_residual = true;
                break;
            }
            const uint32_t oldcount = count;
            count = newcount;
            // Period-2 short cut (exact): the new mask equals the mask two iterations ago, so the loop would
            // alternate between the two masks until max_iter; its outcome is decided by parity alone.
            if (iter >= 1 && !any_differs_prev && newcount >= 3 && iter + 1 < cfg.max_iter) {
                const int remaining = cfg.max_iter - (iter + 1);
                iters = cfg.max_iter;
                if (remaining & 1) {                             // ends on the older mask: pl was fitted on it
                    final_bit = 1; count = oldcount;
                    have_plane = true; list_is_final = false;    // its residual is computed below, over the older mask
                } else {
                    cs[0] = mo.sx; cs[1] = mo.sy; cs[2] = mo.sz;
                }
                break;
            }
            cs[0] = mo.sx; cs[1] = mo.sy; cs[2] = mo.sz;
        }

        // ---- final plane (recursive_patchwork.cpp:220-228) ----------------------------------------
        if (count < 3) {
            pl.cx = pl.cy = pl.cz = 0.0f; pl.nx = 0.0f; pl.ny = 0.0f; pl.nz = 1.0f; pl.residual = 3.402823466e+38f;
        } else {
            if (!have_plane) plane_of(count);                    // cs = sums of the final mask, the list holds it
            if (!have_residual) {
                const bool node_masked = W == 3 && !(kListMaxSlots != 0u && n <= kListMaxSlots);
                if (!list_is_final && !node_masked) {            // nodes with an index list: rebuild it from the older mask
                    (void)w1_mask_pass(nd, kMaskOlder, mp);
                    wave_sync();
                }
                pl.residual = residual_of(count, !list_is_final && node_masked);  // (else: straight from the older mask's plane)
            }
        }

        // ---- split? (recursive_patchwork.cpp:231-305) ---------------------------------------------
        const float split_threshold = cfg.th_dist * (1.0f + 1.5f * (float)depth);
        const uint32_t min_patch = 50u + 10u * (uint32_t)depth;
        if (pl.residual > split_threshold && depth < cfg.max_split_depth && n >= min_patch) {
            const Sums6 m2 = W_SWEEP(nd, n, kSweepXY, 0, no_prm);
            const float fn = (float)n;
            SweepPrm mean = no_prm; mean.a = m2.v[0] / fn; mean.b = m2.v[1] / fn;
            const Sums6 v2 = W_SWEEP(nd, n, kSweepVarXY, 0, mean);
            cx.addends += 2ull * n;
            const float var_x = v2.v[0] / fn, var_y = v2.v[1] / fn;
            const int axis = (var_x > var_y) ? 0 : 1;
            const float median = w1_select_rank(cx, s, n, axis, n / 2);
            const float* src = axis == 0 ? gx : gy;
            // stable partition: left = (val <= median) keeps order, then right keeps order (scratch: the list's array + ay, az)
            uint32_t nleft = 0;
            for (uint32_t b = 0; b < n; b += kWave) {
                const uint32_t i = b + lane;
                nleft += (uint32_t)__popcll(ballot64(i < n && src[i] <= median));
            }
            uint32_t lo = 0, ro = 0;
            for (uint32_t b = 0; b < n; b += kWave) {
                const uint32_t i = b + lane;
                const bool ok = i < n;
                const float val = ok ? src[i] : 0.0f;
                const bool left = ok && val <= median;
                const unsigned long long ml = ballot64(left), mr = ballot64(ok && !left);
                if (ok) {
                    const uint32_t d = cx.base + s + (left ? lo + (uint32_t)mbcnt64(ml) : nleft + ro + (uint32_t)mbcnt64(mr));
                    p.ax[d] = gx[i]; p.ay[d] = gy[i]; p.az[d] = gz[i];
                }
                lo += (uint32_t)__popcll(ml); ro += (uint32_t)__popcll(mr);
            }
            wave_sync();
            for (uint32_t i = lane; i < n; i += kFit1Threads) {
                const uint32_t g = cx.base + s + i;
                p.sx[g] = p.ax[g]; p.sy[g] = p.ay[g]; p.sz[g] = p.az[g];
            }
            wave_sync();
            permuted = true;
            pa.splits++;
            pa.iters += (unsigned long long)iters;
            emit_node(p, lane, frame, patch, depth, s, n, RPW_NODE_SPLIT, iters, count, axis, median, &pl);
            const uint32_t nright = n - nleft;
            if (lane == 0) {
                const bool left_small = nleft <= nright;
                const uint32_t big_s = left_small ? s + nleft : s, big_n = left_small ? nright : nleft;
                const uint32_t sm_s = left_small ? s : s + nleft, sm_n = left_small ? nleft : nright;
                stk[3 * sp] = big_s; stk[3 * sp + 1] = big_n; stk[3 * sp + 2] = (uint32_t)(depth + 1);
                stk[3 * (sp + 1)] = sm_s; stk[3 * (sp + 1) + 1] = sm_n; stk[3 * (sp + 1) + 2] = (uint32_t)(depth + 1);
            }
            sp += 2;
            wave_sync();
            continue;
        }

        // ---- FIT leaf: labels (and residuals) -----------------------------------------------------
        pa.iters += (unsigned long long)iters;
        for (uint32_t i0 = lane; i0 < n; i0 += 4 * kFit1Threads) {
            uint32_t orig[4], m[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + u * kFit1Threads, ic = i < n ? i : n - 1u;
                orig[u] = gi[ic];
                m[u] = (uint32_t)(((final_bit ? mQ : mC)[ic >> 6] >> (ic & 63u)) & 1ull);      // the final mask's plane: the older one after a period-2 ending
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + u * kFit1Threads;
                if (i >= n) continue;
                p.labels[fo + orig[u]] = m[u] ? RPW_LABEL_GROUND : RPW_LABEL_NONGROUND;
                if (p.residual) {
                    float x, y, z;
                    if (permuted) {
                        P3 q = load_point(p, fo + orig[u]);
                        if (p.sensors.n) (void)fuse_point(p, (uint32_t)frame, orig[u], q);
                        x = q.x; y = q.y; z = q.z;
                    }
                    else { x = gx[i]; y = gy[i]; z = gz[i]; }
                    p.residual[fo + orig[u]] = plane_dist(x, y, z, pl.cx, pl.cy, pl.cz, pl.nx, pl.ny, pl.nz);
                }
            }
        }
        pa.ground += count;
        emit_node(p, lane, frame, patch, depth, s, n, RPW_NODE_FIT, iters, count, -1, 0.0f, &pl);
    }

    W_LEAVE()
    if (lane == 0) {
        rpw_frame_stats* st = p.stats + frame;
        atomicAdd((unsigned long long*)&st->chain_addends, cx.addends);
        atomicAdd(&st->n_ground, pa.ground);
        atomicAdd(&st->n_nodes, pa.nodes);
        atomicAdd(&st->n_splits, pa.splits);
        atomicMax(&st->max_depth, pa.maxdepth);
        atomicAdd((unsigned long long*)&st->total_iterations, pa.iters);
    }
}
#undef W1_PH
#undef W_SWEEP
#undef W_FUSED
#undef W_LEAVE

// The single-wave form: one wave per patch runs everything, the dependent adds included (the form the server forms were
// derived from; RPW_FIT_WAVES=1 selects it).
__global__ __launch_bounds__(kFit1Threads, 4) void patch_fit1_kernel(DevParams p)
{
    __shared__ Fit1Shared sh;
    const int lane = threadIdx.x & 63;
    uint32_t linear;
    if (!patch_of_item(p, blockIdx.x, lane, linear)) return;
    fit_patch<1>(p, sh, linear, lane, nullptr, 0);
}

// The chain-server form: wave 0 adds, waves 1..kSrvFeeders take patches from the launch's work counter until it runs out.
__global__ __launch_bounds__(kWave * (1 + kSrvFeeders), 4) void patch_fit_srv_kernel(DevParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char srv_lds[];
    SrvShared* sv = reinterpret_cast<SrvShared*>(srv_lds);
    Fit1Shared* shs = reinterpret_cast<Fit1Shared*>(srv_lds + sizeof(SrvShared));
    srv_init(sv);
    const int lane = threadIdx.x & 63, f = (int)(threadIdx.x >> 6) - 1;
    if (f < 0) { srv_loop(sv, shs); return; }
    for (;;) {
        uint32_t item = 0;
        if (lane == 0) item = atomicAdd(p.work, 1u);
        item = uni(item);
        uint32_t linear;
        if (!patch_of_item(p, item, lane, linear)) break;
        fit_patch<3>(p, shs[f], linear, lane, sv, f);
        wave_sync();
    }
    if (lane == 0 && *(volatile uint32_t*)&sv->error) atomicOr(p.work + 1, 1u);      // a wait timed out: the host reports it
    srv_leave(sv, lane);
}

// The same division of labour for a launch of FEW patches (one frame, a pipeline slot): one workgroup per patch, wave 0 its
// server (srv1_loop), wave 1 the feeder that owns the patch. What counts here is the time of the slowest patch, i.e. cycles
// per addend of one chain: 9.0-9.7 in the sweeps with this server (tools/loop_form_probe.py); round 2's latency form, a chain
// wave that met a helper wave at a barrier per 64 addends, needed 9.8-10.2 and a protocol of its own (removed).
__global__ __launch_bounds__(2 * kWave, 4) void patch_fit_srv1_kernel(DevParams p)
{
    __shared__ SrvShared sv;
    __shared__ Fit1Shared sh;
    __shared__ Srv1Dump dump;
    srv_init(&sv);
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < kWave) { srv1_loop(&sv, &sh, &dump); return; }
    uint32_t linear;
    if (patch_of_item(p, blockIdx.x, lane, linear)) fit_patch<(RPW_SRV1_MASKED ? 3 : 4)>(p, sh, linear, lane, &sv, 0);
    if (lane == 0 && *(volatile uint32_t*)&sv.error) atomicOr(p.work + 1, 1u);      // a wait timed out: the host reports it
    srv_leave(&sv, lane);
}

// ------------------------------------------------------------------------------------------
// debug kernels (test hooks)
// ------------------------------------------------------------------------------------------
__global__ void debug_angles_kernel(const float* x, const float* y, float* a, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = polar_angle(x[i], y[i]);
}
__global__ void debug_eig3_kernel(const float* cov, float* nrm, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float* A = cov + 9 * (size_t)i;
        float nx, ny, nz;
        smallest_eigenvector(A[0], A[3], A[4], A[6], A[7], A[8], nx, ny, nz);
        nrm[3 * (size_t)i] = nx; nrm[3 * (size_t)i + 1] = ny; nrm[3 * (size_t)i + 2] = nz;
    }
}

// ------------------------------------------------------------------------------------------
// micro-benchmark of the two loops patch_fit1 lives in (test hook rpw_debug_loop_bench): every block runs `reps`
// chain sweeps (what = sweep mode) or mask passes (what = 16 + mask mode) over its own synthetic node of n slots
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kFit1Threads, 4) void w1_loop_bench_kernel(const float* sx, const float* sy, const float* sz, unsigned long long* mw, uint32_t mplane,
                                                                            uint32_t* il, uint32_t n, uint32_t count, int what, int reps, float* out, unsigned long long* ticks)
{
    __shared__ Fit1Shared sh;
    const int lane = threadIdx.x & 63;
    lds_f32* ring = (lds_f32*)sh.ring;
    const size_t base = (size_t)blockIdx.x * n;
    NodeRef nd; nd.x = (g_cf32*)(sx + base); nd.y = (g_cf32*)(sy + base); nd.z = (g_cf32*)(sz + base);
    nd.mcur = (g_u64*)mw + (base >> 6) + blockIdx.x; nd.mprev = nd.mcur + mplane; nd.mnew = nd.mprev + mplane; nd.il = (g_u32*)(il + base); nd.n = n;
    float acc = 0.0f;
    const unsigned long long tick0 = __builtin_readcyclecounter();      // shader clock (s_memtime): the chip's clock depends on its recent load
    for (int r = 0; r < reps; ++r) {
        if (what == 6) {                                           // `count` eigenvector solves (one per refit in patch_fit), covariances of a noisy plane
            for (uint32_t i = 0; i < count; ++i) {
                const float t = (float)(i & 15u) * 0.003f + acc * 0.0f;
                const N3 nr = w1_normal(8.1f + t, 0.31f - t, 7.7f, 0.012f + 0.1f * t, -0.02f, 0.0004f + 0.01f * t);
                acc += nr.x;
            }
        } else if (what < 16) {
            SweepPrm prm; prm.a = 0.25f + acc * 0.0f; prm.b = -0.5f; prm.c = 0.125f; prm.d = 0.0f; prm.e = 0.0f; prm.f = 1.0f;
            const Sums6 v = w1_chain_sweep(nd, count, what, 1, prm, ring);
            acc += v.v[0];
        } else {
            MaskPrm mp; mp.cx = 0.0f; mp.cy = 0.0f; mp.cz = acc * 0.0f; mp.nx = 0.0f; mp.ny = 0.0f; mp.nz = 1.0f; mp.thr = 0.6f;
            if (what == 18) { const MaskOut o = w1_mask_pass(nd, kMaskOlder, mp); acc += (float)o.count; }
            else { const FusedOut o = w1_fused_pass(nd, what - 16, mp, ring); acc += (float)o.count + o.sx * 0.0f; }
        }
    }
    if (lane == 0) { out[blockIdx.x] = acc; if (blockIdx.x == 0) ticks[0] = __builtin_readcyclecounter() - tick0; }
}

// the same loops in the chain-server form: workgroup b's feeder f runs them on node b * kSrvFeeders + f
__global__ __launch_bounds__(kWave * (1 + kSrvFeeders), 4) void srv_loop_bench_kernel(const float* sx, const float* sy, const float* sz, unsigned long long* mw, uint32_t mplane,
                                                                                     uint32_t* il, uint32_t n, uint32_t count, int what, int reps, uint32_t nodes, float* out, unsigned long long* ticks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char srv_lds[];
    SrvShared* sv = reinterpret_cast<SrvShared*>(srv_lds);
    Fit1Shared* shs = reinterpret_cast<Fit1Shared*>(srv_lds + sizeof(SrvShared));
    srv_init(sv);
    const int lane = threadIdx.x & 63, f = (int)(threadIdx.x >> 6) - 1;
    if (f < 0) { srv_loop(sv, shs); return; }
    const uint32_t node = blockIdx.x * (uint32_t)kSrvFeeders + (uint32_t)f;
    if (node < nodes) {
        lds_f32* ring = (lds_f32*)shs[f].ring;
        const size_t base = (size_t)node * n;
        NodeRef nd; nd.x = (g_cf32*)(sx + base); nd.y = (g_cf32*)(sy + base); nd.z = (g_cf32*)(sz + base);
        nd.mcur = (g_u64*)mw + (base >> 6) + node; nd.mprev = nd.mcur + mplane; nd.mnew = nd.mprev + mplane; nd.il = (g_u32*)(il + base); nd.n = n;
        float acc = 0.0f;
        const unsigned long long tick0 = __builtin_readcyclecounter();
        for (int r = 0; r < reps; ++r) {
            if (what < 16) {
                SweepPrm prm; prm.a = 0.25f + acc * 0.0f; prm.b = -0.5f; prm.c = 0.125f; prm.d = 0.0f; prm.e = 0.0f; prm.f = 1.0f;
                const Sums6 v = srv_chain_sweep<true>(nd, count, what, prm, ring, sv, f);
                acc += v.v[0];
            } else {
                MaskPrm mp; mp.cx = 0.0f; mp.cy = 0.0f; mp.cz = acc * 0.0f; mp.nx = 0.0f; mp.ny = 0.0f; mp.nz = 1.0f; mp.thr = 0.6f;
                if (what == 18) { const MaskOut o = w1_mask_pass(nd, kMaskOlder, mp); acc += (float)o.count; }
                else { const FusedOut o = srv_fused_pass<false>(nd, what - 16, mp, ring, sv, f); acc += (float)o.count + o.sx * 0.0f; }
            }
        }
        if (lane == 0) { out[node] = acc + (float)sv->error * 1.0e30f; if (node == 0) ticks[0] = __builtin_readcyclecounter() - tick0; }
    }
    srv_leave(sv, lane);
}

// ... and with one server per node: workgroup b = server wave + the feeder of node b
__global__ __launch_bounds__(2 * kWave, 4) void srv1_loop_bench_kernel(const float* sx, const float* sy, const float* sz, unsigned long long* mw, uint32_t mplane,
                                                                      uint32_t* il, uint32_t n, uint32_t count, int what, int reps, float* out, unsigned long long* ticks)
{
    __shared__ SrvShared sv;
    __shared__ Fit1Shared sh;
    __shared__ Srv1Dump dump;
    srv_init(&sv);
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < kWave) { srv1_loop(&sv, &sh, &dump); return; }
    lds_f32* ring = (lds_f32*)sh.ring;
    const size_t base = (size_t)blockIdx.x * n;
    NodeRef nd; nd.x = (g_cf32*)(sx + base); nd.y = (g_cf32*)(sy + base); nd.z = (g_cf32*)(sz + base);
    nd.mcur = (g_u64*)mw + (base >> 6) + blockIdx.x; nd.mprev = nd.mcur + mplane; nd.mnew = nd.mprev + mplane; nd.il = (g_u32*)(il + base); nd.n = n;
    float acc = 0.0f;
    const unsigned long long tick0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (what == 7) {                                           // the server alone: `count` addends published at once (whatever the ring holds)
            const SrvPort pt = srv_port(&sv, 0, ring);
            const uint32_t nslots = (count + kWave - 1) / kWave;
            const uint32_t start = srv_begin(pt, srv_layout((uint32_t)kSlotFloats, (uint32_t)kSrvSweepSlots), 6u, lane);
            uint32_t seen = start;
            for (int i = lane; i < kRingFloats; i += kWave) sh.ring[i] = 1.0f + (float)i * 1.0e-3f;      // ordinary numbers
            wave_sync();
            srv_publish(pt, start + nslots, lane);
            srv_wait(pt, start + nslots, seen);
        } else if (what < 16) {
            SweepPrm prm; prm.a = 0.25f + acc * 0.0f; prm.b = -0.5f; prm.c = 0.125f; prm.d = 0.0f; prm.e = 0.0f; prm.f = 1.0f;
            const Sums6 v = srv_chain_sweep<(RPW_SRV1_MASKED != 0)>(nd, count, what, prm, ring, &sv, 0);
            acc += v.v[0];
        } else {
            MaskPrm mp; mp.cx = 0.0f; mp.cy = 0.0f; mp.cz = acc * 0.0f; mp.nx = 0.0f; mp.ny = 0.0f; mp.nz = 1.0f; mp.thr = 0.6f;
            if (what == 18) { const MaskOut o = w1_mask_pass(nd, kMaskOlder, mp); acc += (float)o.count; }
            else { const FusedOut o = srv_fused_pass<(RPW_SRV1_MASKED == 0)>(nd, what - 16, mp, ring, &sv, 0); acc += (float)o.count + o.sx * 0.0f; }
        }
    }
    if (lane == 0) { out[blockIdx.x] = acc + (float)sv.error * 1.0e30f; if (blockIdx.x == 0) ticks[0] = __builtin_readcyclecounter() - tick0; }
    srv_leave(&sv, lane);
}

// ------------------------------------------------------------------------------------------
// per-frame result record of a batch, as four int32: ground points, non-ground points (the zone's AND the cleaned points
// beyond filtering_radius, which the reference appends to non_ground_points: recursive_patchwork.cpp:415-419), patches, refits
// (what the multi-GPU result gather exchanges: the sizes a reference caller prints, recursive_patchwork.cpp:421-423)
// If patch_fit's error flag is set (a wave timed out waiting for the chain server: the batch's results are invalid), every
// field of every record is -1, so that a consumer that only sees the gathered records cannot mistake them for counts.
__global__ __launch_bounds__(256) void result_record_kernel(const rpw_frame_stats* __restrict__ stats, int32_t* __restrict__ out, uint32_t n_frames,
                                                            const uint32_t* __restrict__ fit_error)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    if (*fit_error) { out[4 * f] = out[4 * f + 1] = out[4 * f + 2] = out[4 * f + 3] = -1; return; }
    const rpw_frame_stats st = stats[f];
    out[4 * f] = (int32_t)st.n_ground; out[4 * f + 1] = (int32_t)(st.n_clean - st.n_ground);
    out[4 * f + 2] = (int32_t)st.n_patches; out[4 * f + 3] = (int32_t)st.total_iterations;
}

// frame offsets / tile bases of a one-frame batch, and the zeroed control words (bucket counts, node count, class totals)
__global__ __launch_bounds__(256) void frame_setup_kernel(uint32_t* offsets, uint32_t n, uint32_t tiles, uint32_t* ctrl, uint32_t ctrl_words)
{
    const uint32_t i = threadIdx.x;
    if (i == 0) { offsets[0] = 0; offsets[1] = n; offsets[2] = 0; offsets[3] = tiles; }
    for (uint32_t j = i; j < ctrl_words; j += blockDim.x) ctrl[j] = 0;
}

// labels (and residuals, and the stats record) of one frame: device memory -> pinned host memory, 16 bytes per lane
__global__ __launch_bounds__(256) void frame_copy_out_kernel(const uint8_t* __restrict__ labels, uint8_t* __restrict__ h_labels, uint32_t n,
                                                             const float* __restrict__ res, float* __restrict__ h_res,
                                                             const rpw_frame_stats* __restrict__ stats, rpw_frame_stats* __restrict__ h_stats,
                                                             const uint32_t* __restrict__ fit_error)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    const uint32_t n16 = n / 16u;
    const uint4* src = reinterpret_cast<const uint4*>(labels);
    uint4* dst = reinterpret_cast<uint4*>(h_labels);
    for (uint32_t i = tid; i < n16; i += nthreads) dst[i] = src[i];
    for (uint32_t i = n16 * 16u + tid; i < n; i += nthreads) h_labels[i] = labels[i];
    if (res) {
        const uint32_t r4 = n / 4u;
        const float4* rs = reinterpret_cast<const float4*>(res);
        float4* rd = reinterpret_cast<float4*>(h_res);
        for (uint32_t i = tid; i < r4; i += nthreads) rd[i] = rs[i];
        for (uint32_t i = r4 * 4u + tid; i < n; i += nthreads) h_res[i] = res[i];
    }
    if (tid == 0) { *h_stats = *stats; *reinterpret_cast<uint32_t*>(h_stats + 1) = *fit_error; }      // (the pinned record has a word behind the stats)
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
#define RPW_LAUNCH_BIN(KERNEL)                                                                              \
    if (p.tile_count == 0) return;                                                                         \
    const size_t lds = (size_t)(4 * p.PB + p.S + 1) * 4;                                                   \
    const dim3 g(p.tile_count), b(kTileThreads);                                                           \
    if (p.sensors.n > 0) {                                                                                 \
        if (p.in_stride == 12u) hipLaunchKernelGGL((KERNEL<false, true>), g, b, lds, s, p);                \
        else                    hipLaunchKernelGGL((KERNEL<true, true>), g, b, lds, s, p);                 \
    } else {                                                                                               \
        if (p.in_stride == 12u) hipLaunchKernelGGL((KERNEL<false, false>), g, b, lds, s, p);               \
        else                    hipLaunchKernelGGL((KERNEL<true, false>), g, b, lds, s, p);                \
    }
void launch_bin_count(const DevParams& p, hipStream_t s) { RPW_LAUNCH_BIN(bin_count_kernel) }
void launch_bin_scatter(const DevParams& p, hipStream_t s) { RPW_LAUNCH_BIN(bin_scatter_kernel) }
#undef RPW_LAUNCH_BIN
void launch_scan(const DevParams& p, hipStream_t s)
{
    if (p.frame_count == 0) return;
    int parts = kScanThreads / (p.P + 4); if (parts > 16) parts = 16; if (parts < 1) parts = 1;      // as in the kernel
    hipLaunchKernelGGL(scan_offsets_kernel, dim3(p.frame_count), dim3(kScanThreads), (size_t)p.PB * (2 + parts) * 4, s, p);
}
// The chain-server workgroup needs more than the default 64 KB of LDS: the opt-in is per kernel AND per device (a process
// may hold handles on several GPUs), so it is made once for every (kernel, current device) pair.
static void allow_srv_lds(const void* kernel)
{
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    if (done.insert({kernel, dev}).second)
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSrvLdsBytes);
}

void launch_fit(const DevParams& p, hipStream_t s)
{
    const uint32_t grid = p.n_frames * (uint32_t)p.P;
    if (grid == 0) return;
    // RPW_FIT_MAXWG=k (experiments): at most k workgroups of patch_fit1 per CU, by padding each with unused dynamic LDS
    static const size_t pad = [] {
        const char* e = getenv("RPW_FIT_MAXWG");
        const int k = e ? atoi(e) : 0;
        if (k <= 0 || k >= 16) return (size_t)0;
        const size_t per = (160u * 1024u) / (size_t)k, stat = sizeof(Fit1Shared) + 64;
        return per > stat ? ((per - stat) & ~(size_t)255) : (size_t)0;
    }();
    // RPW_FIT_WAVES=1|3|4 forces a form (tests, A/B measurements): one wave per patch, one chain server for seven patches, one
    // server per patch.
    static const int forced = [] { const char* e = getenv("RPW_FIT_WAVES"); return e ? atoi(e) : 0; }();
    if (forced == 1) { hipLaunchKernelGGL(patch_fit1_kernel, dim3(grid), dim3(kFit1Threads), pad, s, p); return; }
    // few patches in the launch (a frame or a few: the host-pointer path) -> a server wave of its own for every patch, the
    // form with the shortest time per patch; a batch -> one server for seven patch owners, the form with the fewest
    // instructions per point
    if (forced == 4 || (forced == 0 && grid <= 8u * 256u)) { hipLaunchKernelGGL(patch_fit_srv1_kernel, dim3(grid), dim3(2 * kWave), 0, s, p); return; }
    // chain-server form: persistent workgroups (two fit a CU), each feeder pulls patches until the work counter runs out
    allow_srv_lds(reinterpret_cast<const void*>(patch_fit_srv_kernel));
    uint32_t wgs = (grid + (uint32_t)kSrvFeeders - 1u) / (uint32_t)kSrvFeeders;
    // RPW_SRV_WGS=k (experiments): fewer persistent workgroups than fit the chip, i.e. fewer patches in flight
    static const uint32_t cap = [] { const char* e = getenv("RPW_SRV_WGS"); const int k = e ? atoi(e) : 0; return (k > 0 && k < 512) ? (uint32_t)k : 2u * 256u; }();
    if (wgs > cap) wgs = cap;
    hipLaunchKernelGGL(patch_fit_srv_kernel, dim3(wgs), dim3(kWave * (1 + kSrvFeeders)), kSrvLdsBytes, s, p);
}
void launch_result_record(const rpw_frame_stats* stats, int32_t* out, uint32_t n_frames, const uint32_t* fit_error, hipStream_t s)
{
    if (n_frames) hipLaunchKernelGGL(result_record_kernel, dim3((n_frames + 255) / 256), dim3(256), 0, s, stats, out, n_frames, fit_error);
}
void launch_frame_setup(uint32_t* offsets, uint32_t n, uint32_t tiles, uint32_t* ctrl, uint32_t ctrl_words, hipStream_t s)
{
    hipLaunchKernelGGL(frame_setup_kernel, dim3(1), dim3(256), 0, s, offsets, n, tiles, ctrl, ctrl_words);
}
void launch_frame_copy_out(const uint8_t* labels, uint8_t* h_labels, uint32_t n, const float* res, float* h_res,
                           const rpw_frame_stats* stats, rpw_frame_stats* h_stats, const uint32_t* fit_error, hipStream_t s)
{
    uint32_t blocks = (n / 16u + 255u) / 256u;
    if (blocks < 1) blocks = 1;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(frame_copy_out_kernel, dim3(blocks), dim3(256), 0, s, labels, h_labels, n, res, h_res, stats, h_stats, fit_error);
}
void launch_loop_bench(const float* sx, const float* sy, const float* sz, unsigned long long* mw, uint32_t mplane, uint32_t* il, uint32_t n, uint32_t count,
                       int what, int reps, uint32_t blocks, float* out, unsigned long long* ticks, hipStream_t s)
{
    const int w = 1 + (what >> 5);      // what + 64: the same loop in the chain-server form (seven nodes per server), + 96: one server per node
    what &= 31;
    if (w == 4) { hipLaunchKernelGGL(srv1_loop_bench_kernel, dim3(blocks), dim3(2 * kWave), 0, s, sx, sy, sz, mw, mplane, il, n, count, what, reps, out, ticks); return; }
    if (w == 3) {
        allow_srv_lds(reinterpret_cast<const void*>(srv_loop_bench_kernel));
        const uint32_t wgs = (blocks + (uint32_t)kSrvFeeders - 1u) / (uint32_t)kSrvFeeders;
        hipLaunchKernelGGL(srv_loop_bench_kernel, dim3(wgs), dim3(kWave * (1 + kSrvFeeders)), kSrvLdsBytes, s, sx, sy, sz, mw, mplane, il, n, count, what, reps, blocks, out, ticks);
        return;
    }
    hipLaunchKernelGGL(w1_loop_bench_kernel, dim3(blocks), dim3(kFit1Threads), 0, s, sx, sy, sz, mw, mplane, il, n, count, what, reps, out, ticks);
}
void launch_debug_angles(const float* x, const float* y, float* a, uint32_t n, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(debug_angles_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, y, a, n);
}
void launch_debug_eig3(const float* cov9, float* normal3, uint32_t n, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(debug_eig3_kernel, dim3((n + 255) / 256), dim3(256), 0, s, cov9, normal3, n);
}

} // namespace rpw
