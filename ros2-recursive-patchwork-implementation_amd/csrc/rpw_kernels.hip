// rpw_kernels.hip -- hand-written gfx950 kernels of the Recursive Patchwork hot path.
//
//   K1 classify_hist   cleanPoints + computeDistances2D + filterPointsByRadius + computeAngles +
//                      patch membership (recursive_patchwork.cpp:315-378), O(N) instead of O(P*N)
//   K2 scan_offsets    per-frame exclusive offsets (patch-major, then tile order = input order)
//   K3 stable_scatter  the gather of recursive_patchwork.cpp:372-378 for all patches at once:
//                      patch-sorted SoA x,y,z + original index, input order kept inside a patch
//   K4 patch_fit       fitPlaneAndSplit / fitPlanePCA / computeCentroid / computeCovariance /
//                      Eigen solve / computePlaneDistances / mask compare / split + recursion
//                      (recursive_patchwork.cpp:77-308), one wavefront per (frame, patch)
//
// Exactness: every sum the reference accumulates sequentially in binary32 (centroid,
// covariance, residual, mean distance, variance) is accumulated here in the same order
// by one lane per accumulator ("chain lanes"); the other 63 lanes of the wavefront only
// prepare the addends. Built with -ffp-contract=off; divide and sqrt are correctly rounded.
#include <hip/hip_runtime.h>
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
__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ float uniform_f32(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// ------------------------------------------------------------------------------------------
// K1: classify + per-tile histogram
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t frame_of_tile(const uint32_t* __restrict__ tile_base, uint32_t n_frames, uint32_t tile)
{
    uint32_t lo = 0, hi = n_frames;          // tile_base[lo] <= tile < tile_base[hi]
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (tile_base[mid] <= tile) lo = mid; else hi = mid; }
    return lo;
}

__global__ __launch_bounds__(kTileThreads) void classify_hist_kernel(DevParams p)
{
    extern __shared__ uint32_t lds_u32[];
    uint32_t* hist = lds_u32;                              // [PB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int j = tid; j < p.PB; j += kTileThreads) hist[j] = 0;
    __syncthreads();

    const uint32_t tile = blockIdx.x;
    const uint32_t f = frame_of_tile(p.tile_base, p.n_frames, tile);
    const uint32_t fo = p.frame_off[f];
    const uint32_t nf = p.frame_off[f + 1] - fo;
    const uint32_t t_local = tile - p.tile_base[f];
    const uint32_t wbase = t_local * kTilePoints + wave * kWavePoints;

#pragma unroll 2
    for (int r = 0; r < kTileItems; ++r) {
        const uint32_t i = wbase + r * kWave + lane;
        const bool valid = i < nf;
        int cls = 0;
        if (valid) {
            const P3 q = p.xyz[fo + i];
            cls = classify_point(q.x, q.y, q.z, p.tab, p.bounds);
            p.pid[fo + i] = (uint16_t)cls;
        }
        unsigned long long rem = ballot64(valid);
        while (rem) {
            const int first = __ffsll((long long)rem) - 1;
            const int c0 = __shfl(cls, first);
            const unsigned long long m = ballot64(valid && cls == c0);
            if (lane == first) atomicAdd(&hist[c0], (uint32_t)__popcll(m));
            rem &= ~m;
        }
    }
    __syncthreads();
    uint32_t* out = p.hist + (size_t)tile * p.PB;
    for (int j = tid; j < p.PB; j += kTileThreads) out[j] = hist[j];
}

// ------------------------------------------------------------------------------------------
// K2: per-frame offsets. One 256-thread block per frame.
// ------------------------------------------------------------------------------------------
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

__device__ __forceinline__ int size_bucket(uint32_t n)
{
    // two buckets per power of two: 2*floor(log2 n) + next bit  (n >= 1)
    const int lg = 31 - __clz((int)n);
    const int half = lg > 0 ? (int)((n >> (lg - 1)) & 1u) : 0;
    int b = 2 * lg + half;
    return b < kBuckets ? b : kBuckets - 1;
}

__global__ __launch_bounds__(kTileThreads) void scan_offsets_kernel(DevParams p)
{
    __shared__ uint32_t red[8];
    const int tid = threadIdx.x;
    const uint32_t f = blockIdx.x;
    const uint32_t t0 = p.tile_base[f];
    const uint32_t nt = p.tile_base[f + 1] - t0;
    const uint32_t fo = p.frame_off[f];
    const uint32_t nf = p.frame_off[f + 1] - fo;
    uint32_t running = 0, npatches = 0;
    for (int c = 0; c < p.P; ++c) {
        uint32_t carry = 0;
        for (uint32_t b0 = 0; b0 < nt; b0 += kTileThreads) {
            const uint32_t t = b0 + tid;
            uint32_t* cell = p.hist + (size_t)(t0 + t) * p.PB + c;
            const uint32_t v = t < nt ? *cell : 0u;
            uint32_t total;
            const uint32_t ex = block_excl_scan(v, red, total);
            if (t < nt) *cell = running + carry + ex;
            carry += total;
        }
        if (tid == 0) {
            p.patch_begin[(size_t)f * (p.P + 1) + c] = running;
            PatchDesc d; d.begin = fo + running; d.n = carry;
            p.patches[(size_t)f * p.P + c] = d;
            if (carry > 0) {
                const int b = size_bucket(carry);
                const uint32_t pos = atomicAdd(&p.bucket_count[b], 1u);
                p.bucket_list[(size_t)b * p.list_stride + pos] = f * (uint32_t)p.P + (uint32_t)c;
            }
        }
        if (carry > 0) ++npatches;
        running += carry;
    }
    // special classes: totals only
    uint32_t special[3];
    for (int c = 0; c < 3; ++c) {
        uint32_t acc = 0;
        for (uint32_t b0 = 0; b0 < nt; b0 += kTileThreads) {
            const uint32_t t = b0 + tid;
            const uint32_t v = t < nt ? p.hist[(size_t)(t0 + t) * p.PB + p.P + c] : 0u;
            uint32_t total;
            (void)block_excl_scan(v, red, total);
            acc += total;
        }
        special[c] = acc;
    }
    if (tid == 0) {
        p.patch_begin[(size_t)f * (p.P + 1) + p.P] = running;
        rpw_frame_stats st;
        st.n_in = nf;
        st.n_clean = nf - special[2];
        st.n_zone = st.n_clean - special[1];
        st.n_ground = 0; st.n_patches = npatches; st.n_nodes = 0; st.n_splits = 0; st.max_depth = 0;
        st.total_iterations = 0;
        p.stats[f] = st;
    }
}

// ------------------------------------------------------------------------------------------
// K3: stable scatter to patch-sorted SoA
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kTileThreads) void stable_scatter_kernel(DevParams p)
{
    extern __shared__ uint32_t lds_u32[];
    volatile uint32_t* wcnt = lds_u32;                    // [4][PB] per-wave running counts
    uint32_t* toff = lds_u32 + 4 * p.PB;                  // [PB] tile offsets (frame-relative)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tile = blockIdx.x;
    for (int j = tid; j < 4 * p.PB; j += kTileThreads) wcnt[j] = 0;
    for (int j = tid; j < p.PB; j += kTileThreads) toff[j] = p.hist[(size_t)tile * p.PB + j];
    __syncthreads();

    const uint32_t f = frame_of_tile(p.tile_base, p.n_frames, tile);
    const uint32_t fo = p.frame_off[f];
    const uint32_t nf = p.frame_off[f + 1] - fo;
    const uint32_t t_local = tile - p.tile_base[f];
    const uint32_t wbase = t_local * kTilePoints + wave * kWavePoints;
    volatile uint32_t* mycnt = wcnt + wave * p.PB;

    float px[kTileItems], py[kTileItems], pz[kTileItems];
    int cls[kTileItems];
    uint32_t loff[kTileItems];
#pragma unroll
    for (int r = 0; r < kTileItems; ++r) {
        const uint32_t i = wbase + r * kWave + lane;
        const bool valid = i < nf;
        cls[r] = p.P;  loff[r] = 0; px[r] = py[r] = pz[r] = 0.0f;
        if (valid) {
            const P3 q = p.xyz[fo + i];
            px[r] = q.x; py[r] = q.y; pz[r] = q.z;
            cls[r] = p.pid[fo + i];
            const int c = cls[r];
            p.labels[fo + i] = (uint8_t)(c < p.P ? RPW_LABEL_NONGROUND
                                       : (c == p.P ? RPW_LABEL_NONGROUND : (c == p.P + 1 ? RPW_LABEL_OUTSIDE : RPW_LABEL_DROPPED)));
            if (p.residual) p.residual[fo + i] = -1.0f;
        }
        const bool inpatch = valid && cls[r] < p.P;
        unsigned long long rem = ballot64(inpatch);
        while (rem) {
            const int first = __ffsll((long long)rem) - 1;
            const int c0 = __shfl(cls[r], first);
            const unsigned long long m = ballot64(inpatch && cls[r] == c0);
            const uint32_t old = mycnt[c0];
            if (inpatch && cls[r] == c0) loff[r] = old + (uint32_t)mbcnt64(m);
            if (lane == first) mycnt[c0] = old + (uint32_t)__popcll(m);
            rem &= ~m;
        }
    }
    __syncthreads();
    for (int j = tid; j < p.P; j += kTileThreads) {       // exclusive prefix over the 4 waves
        uint32_t s = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const uint32_t t = wcnt[w * p.PB + j]; wcnt[w * p.PB + j] = s; s += t; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kTileItems; ++r) {
        const uint32_t i = wbase + r * kWave + lane;
        if (i < nf && cls[r] < p.P) {
            const uint32_t dest = fo + toff[cls[r]] + mycnt[cls[r]] + loff[r];
            p.sx[dest] = px[r]; p.sy[dest] = py[r]; p.sz[dest] = pz[r];
            p.sidx[dest] = i;
        }
    }
}

// ------------------------------------------------------------------------------------------
// K4: per-patch recursive plane fit, one wavefront per (frame, patch)
// ------------------------------------------------------------------------------------------
constexpr int kLdw = 68;   // staging row stride in floats: 64 + 4 keeps the 6 chain lanes on distinct banks

struct Raw { float x, y, z; uint32_t m; };

struct FitCtx {
    const DevParams* p;
    uint32_t base;        // absolute slot of the patch's first point
    float* stage;         // [6][kLdw]
    int lane;
};

__device__ __forceinline__ Raw load_raw(const FitCtx& cx, uint32_t s, uint32_t n, uint32_t chunk)
{
    Raw r; r.x = 0.0f; r.y = 0.0f; r.z = 0.0f; r.m = 0u;
    const uint32_t i = chunk * kWave + cx.lane;
    if (i < n) {
        const uint32_t g = cx.base + s + i;
        r.x = cx.p->sx[g]; r.y = cx.p->sy[g]; r.z = cx.p->sz[g];
        r.m = (uint32_t)cx.p->mask[g] | 0x100u;          // bit 8 = slot exists
    }
    return r;
}

// 64 sequential binary32 adds per chain lane; rows hold the addends in slot order, padded with -0.0f
// (x + -0.0f == x for every x, so padding and masked-out slots are exact no-ops).
__device__ __forceinline__ void chain64(float& acc, const float* row)
{
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float4 q = *reinterpret_cast<const float4*>(row + 4 * j);
        acc += q.x; acc += q.y; acc += q.z; acc += q.w;
    }
}

// Generic chained pass over a node: K accumulators live in lanes 0..K-1.
// make(raw, out[K]) turns one slot into its K addends.
template <int K, class Make>
__device__ __forceinline__ float chain_pass(const FitCtx& cx, uint32_t s, uint32_t n, Make make)
{
    float acc = 0.0f;                                     // Zero(), then += in slot order
    const uint32_t nchunks = (n + kWave - 1) / kWave;
    const int row = cx.lane < K ? cx.lane : K - 1;
    Raw nxt = load_raw(cx, s, n, 0);
    for (uint32_t c = 0; c < nchunks; ++c) {
        const Raw cur = nxt;
        if (c + 1 < nchunks) nxt = load_raw(cx, s, n, c + 1);
        float v[K];
        make(cur, v);
#pragma unroll
        for (int k = 0; k < K; ++k) cx.stage[k * kLdw + cx.lane] = v[k];
        __syncthreads();
        chain64(acc, cx.stage + row * kLdw);
        __syncthreads();
    }
    return acc;                                           // lane k < K holds accumulator k
}

__device__ __forceinline__ float plane_dist(float x, float y, float z, float cx_, float cy_, float cz_, float nx, float ny, float nz)
{
    const float e0 = (x - cx_) * nx;
    const float e1 = (y - cy_) * ny;
    const float e2 = (z - cz_) * nz;
    return fabs_(e0 + (e1 + e2));
}

struct PlaneD { float cx, cy, cz, nx, ny, nz, residual; };

// centroid + covariance + eigenvector over the slots whose mask bit `bit` is set (m = their count >= 3)
__device__ __forceinline__ void fit_plane_masked(const FitCtx& cx, uint32_t s, uint32_t n, int bit, uint32_t m, PlaneD& pl)
{
    const float NZ = -0.0f;
    float acc = chain_pass<3>(cx, s, n, [&](const Raw& r, float* v) {
        const bool in = (r.m >> bit) & 1u;
        v[0] = in ? r.x : NZ; v[1] = in ? r.y : NZ; v[2] = in ? r.z : NZ;
    });
    const float fm = (float)m;
    const float c0 = __shfl(acc, 0) / fm, c1 = __shfl(acc, 1) / fm, c2 = __shfl(acc, 2) / fm;
    acc = chain_pass<6>(cx, s, n, [&](const Raw& r, float* v) {
        const bool in = (r.m >> bit) & 1u;
        const float dx = r.x - c0, dy = r.y - c1, dz = r.z - c2;
        v[0] = in ? dx * dx : NZ;   // (0,0)
        v[1] = in ? dy * dx : NZ;   // (1,0)
        v[2] = in ? dy * dy : NZ;   // (1,1)
        v[3] = in ? dz * dx : NZ;   // (2,0)
        v[4] = in ? dz * dy : NZ;   // (2,1)
        v[5] = in ? dz * dz : NZ;   // (2,2)
    });
    const float fm1 = (float)(m - 1);
    const float a00 = __shfl(acc, 0) / fm1, a10 = __shfl(acc, 1) / fm1, a11 = __shfl(acc, 2) / fm1;
    const float a20 = __shfl(acc, 3) / fm1, a21 = __shfl(acc, 4) / fm1, a22 = __shfl(acc, 5) / fm1;
    pl.cx = c0; pl.cy = c1; pl.cz = c2;
    smallest_eigenvector(a00, a10, a11, a20, a21, a22, pl.nx, pl.ny, pl.nz);
}

__device__ __forceinline__ float residual_masked(const FitCtx& cx, uint32_t s, uint32_t n, int bit, uint32_t m, const PlaneD& pl)
{
    const float acc = chain_pass<1>(cx, s, n, [&](const Raw& r, float* v) {
        const bool in = (r.m >> bit) & 1u;
        v[0] = in ? plane_dist(r.x, r.y, r.z, pl.cx, pl.cy, pl.cz, pl.nx, pl.ny, pl.nz) : -0.0f;
    });
    return __shfl(acc, 0) / (float)m;
}

// monotone key: ascending unsigned order == ascending float order
__device__ __forceinline__ uint32_t float_key(float v) { const uint32_t u = __float_as_uint(v); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float key_float(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// exact order statistic: the value of rank k (0-based, ascending) among coordinate `axis` of the node
__device__ float select_rank(const FitCtx& cx, uint32_t s, uint32_t n, int axis, uint32_t k, uint32_t* hist /*[256] LDS*/)
{
    const float* src = axis == 0 ? cx.p->sx : (axis == 1 ? cx.p->sy : cx.p->sz);
    uint32_t prefix = 0, pmask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int j = cx.lane; j < 256; j += kWave) hist[j] = 0;
        __syncthreads();
        for (uint32_t i = cx.lane; i < n; i += kWave) {
            const uint32_t key = float_key(src[cx.base + s + i]);
            if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        // lane l owns bins 4l..4l+3
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
        __syncthreads();
    }
    return key_float(prefix);
}

// libstdc++ std::partial_sort(idx, idx+3, idx+n, z[a] < z[b]) restated for "the 3 lowest z":
// __heap_select over a 3-element max-heap; returns the three selected slots (set semantics).
__device__ void three_lowest(const FitCtx& cx, uint32_t s, uint32_t n, uint32_t sel[3])
{
    const float* zs = cx.p->sz + cx.base + s;
    uint32_t h0 = 0, h1 = 1, h2 = 2;
    float z0 = zs[0], z1 = zs[1], z2 = zs[2];
    // __adjust_heap(first, 0, 3, value) + __push_heap, value = (vi, vz)
    auto adjust = [&](uint32_t vi, float vz) {
        // larger child moves up into the hole at the root
        int hole;
        if (z2 < z1) { h0 = h1; z0 = z1; hole = 1; } else { h0 = h2; z0 = z2; hole = 2; }
        // __push_heap: parent of hole is the root
        if (z0 < vz) { if (hole == 1) { h1 = h0; z1 = z0; } else { h2 = h0; z2 = z0; } h0 = vi; z0 = vz; }
        else { if (hole == 1) { h1 = vi; z1 = vz; } else { h2 = vi; z2 = vz; } }
    };
    adjust(h0, z0);                                       // __make_heap: parent = 0, value = first[0]
    for (uint32_t b = 3; b < n; b += kWave) {             // for (i = middle; i < last; ++i) if (z[i] < z[top]) pop_heap
        const uint32_t i = b + cx.lane;
        const float zi = i < n ? zs[i] : 0.0f;
        unsigned long long done = 0ull;
        for (;;) {
            const bool want = i < n && zi < z0 && !((done >> cx.lane) & 1ull) ;
            // lanes already passed (below the current position) must not be revisited: only lanes >= cursor
            unsigned long long cand = ballot64(want);
            if (!cand) break;
            const int l = __ffsll((long long)cand) - 1;
            const uint32_t vi = (uint32_t)__shfl((int)i, l);
            const float vz = __shfl(zi, l);
            adjust(vi, vz);
            // every lane up to and including l has now been consumed in order
            done |= (l == 63) ? ~0ull : ((1ull << (l + 1)) - 1ull);
        }
    }
    sel[0] = h0; sel[1] = h1; sel[2] = h2;
}

struct PatchAcc { uint32_t nodes, splits, maxdepth, ground; unsigned long long iters; };

__device__ __forceinline__ void emit_node(const DevParams& p, int lane, int frame, int patch, int depth, uint32_t start, uint32_t len,
                                          int kind, int iters, uint32_t ninl, int axis, float sval, const PlaneD* pl)
{
    if (!p.trace || lane != 0) return;
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

__global__ __launch_bounds__(kWave) void patch_fit_kernel(DevParams p)
{
    __shared__ __attribute__((aligned(16))) float stage[6 * kLdw];
    __shared__ uint32_t hist[256];
    __shared__ uint32_t stk[3 * kStackDepth];
    const int lane = threadIdx.x;

    // ---- block -> patch, largest size bucket first --------------------------------------------
    uint32_t linear;
    {
        const uint32_t cnt = p.bucket_count[kBuckets - 1 - lane];      // lane l: bucket 63-l (descending size)
        const uint32_t incl = wave_incl_scan(cnt);
        const uint32_t total = __shfl(incl, 63);
        if (blockIdx.x >= total) return;
        const unsigned long long bm = ballot64(blockIdx.x < incl);
        const int l = __ffsll((long long)bm) - 1;
        const uint32_t before = __shfl(incl - cnt, l);
        linear = p.bucket_list[(size_t)(kBuckets - 1 - l) * p.list_stride + (blockIdx.x - before)];
    }
    const int frame = (int)(linear / (uint32_t)p.P), patch = (int)(linear % (uint32_t)p.P);
    const PatchDesc pd = p.patches[linear];
    const uint32_t fo = p.frame_off[frame];

    FitCtx cx; cx.p = &p; cx.base = pd.begin; cx.stage = stage; cx.lane = lane;
    const rpw_config& cfg = p.cfg;

    // ---- mean distance of the patch (recursive_patchwork.cpp:383-387): sequential sum of d ------
    float mean_dist;
    {
        const float acc = chain_pass<1>(cx, 0, pd.n, [&](const Raw& r, float* v) {
            v[0] = (r.m & 0x100u) ? sqrt_rn(r.x * r.x + r.y * r.y) : -0.0f;
        });
        mean_dist = __shfl(acc, 0) / (float)pd.n;
    }
    const float rel_dist = mean_dist / cfg.filtering_radius;
    const float threshold = cfg.th_dist * (1.0f + 0.2f * rel_dist);

    PatchAcc pa; pa.nodes = 0; pa.splits = 0; pa.maxdepth = 0; pa.ground = 0; pa.iters = 0;
    bool permuted = false;

    int sp = 0;
    if (lane == 0) { stk[0] = 0; stk[1] = pd.n; stk[2] = 0; }
    sp = 1;
    __syncthreads();

    while (sp > 0) {
        --sp;
        const uint32_t s = stk[3 * sp], n = stk[3 * sp + 1];
        const int depth = (int)stk[3 * sp + 2];
        __syncthreads();
        pa.nodes++;
        if ((uint32_t)depth > pa.maxdepth) pa.maxdepth = (uint32_t)depth;

        // ---- early outs (recursive_patchwork.cpp:111-140) ---------------------------------------
        int leaf_kind = -1;            // -1: run the fit
        if (n < 3 || depth > cfg.max_split_depth) leaf_kind = RPW_NODE_REJECT;
        float zmin_all = 0.0f;
        if (leaf_kind < 0) {
            float xmn = 3.402823466e+38f, xmx = -3.402823466e+38f, ymn = xmn, ymx = xmx, zmn = xmn, zmx = xmx;
            for (uint32_t i = lane; i < n; i += kWave) {
                const uint32_t g = cx.base + s + i;
                const float x = p.sx[g], y = p.sy[g], z = p.sz[g];
                xmn = fminf(xmn, x); xmx = fmaxf(xmx, x); ymn = fminf(ymn, y); ymx = fmaxf(ymx, y);
                zmn = fminf(zmn, z); zmx = fmaxf(zmx, z);
            }
            xmn = wave_min(xmn); xmx = wave_max(xmx); ymn = wave_min(ymn); ymx = wave_max(ymx);
            zmn = wave_min(zmn); zmx = wave_max(zmx);
            zmin_all = zmn;
            const float area = (xmx - xmn) * (ymx - ymn);
            if (area < 25.0f && depth > 0) leaf_kind = RPW_NODE_SMALL;
            else if ((zmx - zmn) < 0.05f && n > 10) leaf_kind = RPW_NODE_FLAT;
        }
        (void)zmin_all;
        if (leaf_kind >= 0) {
            if (leaf_kind != RPW_NODE_REJECT) {
                for (uint32_t i = lane; i < n; i += kWave)
                    p.labels[fo + p.sidx[cx.base + s + i]] = RPW_LABEL_GROUND;
                pa.ground += n;
            }
            emit_node(p, lane, frame, patch, depth, s, n, leaf_kind, 0, leaf_kind == RPW_NODE_REJECT ? 0u : n, -1, 0.0f, nullptr);
            continue;
        }

        // ---- seeds (recursive_patchwork.cpp:149-182) ---------------------------------------------
        float z_th;
        if (cfg.adaptive_seed_height) {
            z_th = cfg.sensor_height + 0.2f * rel_dist;
        } else {
            const uint32_t idx = (uint32_t)(0.1f * (float)n);     // static_cast<size_t>(0.1f * size)
            z_th = select_rank(cx, s, n, 2, idx < n ? idx : n - 1, hist) + cfg.th_seeds;
        }
        uint32_t count = 0;
        for (uint32_t b = 0; b < n; b += kWave) {
            const uint32_t i = b + lane;
            const bool seed = i < n && p.sz[cx.base + s + i] < z_th;
            if (i < n) p.mask[cx.base + s + i] = seed ? 1 : 0;
            count += (uint32_t)__popcll(ballot64(seed));
        }
        __syncthreads();
        if (count < 3) {
            uint32_t sel[3];
            three_lowest(cx, s, n, sel);
            for (uint32_t i = lane; i < n; i += kWave)
                p.mask[cx.base + s + i] = (i == sel[0] || i == sel[1] || i == sel[2]) ? 1 : 0;
            count = 3;
            __syncthreads();
        }

        // ---- iterative refit (recursive_patchwork.cpp:186-217) ------------------------------------
        PlaneD pl; pl.cx = pl.cy = pl.cz = 0.0f; pl.nx = 0.0f; pl.ny = 0.0f; pl.nz = 1.0f; pl.residual = 3.402823466e+38f;
        int iters = 0;
        int final_bit = 0;
        bool plane_is_final = false;       // pl was fitted on exactly the final mask
        for (int iter = 0; iter < cfg.max_iter; ++iter) {
            if (count < 3) break;
            fit_plane_masked(cx, s, n, 0, count, pl);
            ++iters;
            // new mask = d < threshold; compare with current (bit0) and with the one before (bit1)
            bool changed = false, differs_prev = false;
            uint32_t newcount = 0;
            for (uint32_t b = 0; b < n; b += kWave) {
                const uint32_t i = b + lane;
                bool nw = false;
                if (i < n) {
                    const uint32_t g = cx.base + s + i;
                    const uint32_t mk = p.mask[g];
                    nw = plane_dist(p.sx[g], p.sy[g], p.sz[g], pl.cx, pl.cy, pl.cz, pl.nx, pl.ny, pl.nz) < threshold;
                    const uint32_t old = mk & 1u, prev = (mk >> 1) & 1u;
                    changed |= ((uint32_t)nw != old);
                    differs_prev |= ((uint32_t)nw != prev);
                    p.mask[g] = (uint8_t)((uint32_t)nw | (old << 1));
                }
                newcount += (uint32_t)__popcll(ballot64(nw));
            }
            const bool any_changed = ballot64(changed) != 0ull;
            const bool any_differs_prev = ballot64(differs_prev) != 0ull;
            __syncthreads();
            if (!any_changed) { plane_is_final = true; break; }       // new_mask == ground_mask
            const uint32_t oldcount = count;
            count = newcount;
            plane_is_final = false;
            // Period-2 short cut: the new mask equals the mask two iterations ago, so the loop would
            // alternate between the two masks until max_iter; its outcome is decided by parity alone.
            if (iter >= 1 && !any_differs_prev && newcount >= 3 && iter + 1 < cfg.max_iter) {
                const int remaining = cfg.max_iter - (iter + 1);
                iters = cfg.max_iter;
                if (remaining & 1) { final_bit = 1; count = oldcount; plane_is_final = true; }   // ends on the older mask: pl was fitted on it
                break;
            }
        }

        // ---- final plane (recursive_patchwork.cpp:220-228) ----------------------------------------
        if (count < 3) {
            pl.cx = pl.cy = pl.cz = 0.0f; pl.nx = 0.0f; pl.ny = 0.0f; pl.nz = 1.0f; pl.residual = 3.402823466e+38f;
        } else {
            if (!plane_is_final) fit_plane_masked(cx, s, n, final_bit, count, pl);
            pl.residual = residual_masked(cx, s, n, final_bit, count, pl);
        }

        // ---- split? (recursive_patchwork.cpp:231-305) ---------------------------------------------
        const float split_threshold = cfg.th_dist * (1.0f + 1.5f * (float)depth);
        const uint32_t min_patch = 50u + 10u * (uint32_t)depth;
        if (pl.residual > split_threshold && depth < cfg.max_split_depth && n >= min_patch) {
            // variance of x and y about the centroid of ALL node points, sequential sums
            float acc = chain_pass<2>(cx, s, n, [&](const Raw& r, float* v) {
                const bool ok = r.m & 0x100u;
                v[0] = ok ? r.x : -0.0f; v[1] = ok ? r.y : -0.0f;
            });
            const float fn = (float)n;
            const float mx = __shfl(acc, 0) / fn, my = __shfl(acc, 1) / fn;
            acc = chain_pass<2>(cx, s, n, [&](const Raw& r, float* v) {
                const bool ok = r.m & 0x100u;
                const float dx = r.x - mx, dy = r.y - my;
                v[0] = ok ? dx * dx : -0.0f; v[1] = ok ? dy * dy : -0.0f;
            });
            const float var_x = __shfl(acc, 0) / fn, var_y = __shfl(acc, 1) / fn;
            const int axis = (var_x > var_y) ? 0 : 1;
            const float median = select_rank(cx, s, n, axis, n / 2, hist);
            const float* src = axis == 0 ? p.sx : p.sy;
            // stable partition: left = (val <= median) keeps order, then right keeps order
            uint32_t nleft = 0;
            for (uint32_t b = 0; b < n; b += kWave) {
                const uint32_t i = b + lane;
                nleft += (uint32_t)__popcll(ballot64(i < n && src[cx.base + s + i] <= median));
            }
            uint32_t lo = 0, ro = 0;
            for (uint32_t b = 0; b < n; b += kWave) {
                const uint32_t i = b + lane;
                const bool ok = i < n;
                float x = 0, y = 0, z = 0;
                if (ok) { const uint32_t g = cx.base + s + i; x = p.sx[g]; y = p.sy[g]; z = p.sz[g]; }
                const bool left = ok && (axis == 0 ? x : y) <= median;
                const unsigned long long ml = ballot64(left), mr = ballot64(ok && !left);
                if (ok) {
                    const uint32_t d = cx.base + s + (left ? lo + (uint32_t)mbcnt64(ml) : nleft + ro + (uint32_t)mbcnt64(mr));
                    p.ax[d] = x; p.ay[d] = y; p.az[d] = z;
                }
                lo += (uint32_t)__popcll(ml); ro += (uint32_t)__popcll(mr);
            }
            __syncthreads();
            for (uint32_t i = lane; i < n; i += kWave) {
                const uint32_t g = cx.base + s + i;
                p.sx[g] = p.ax[g]; p.sy[g] = p.ay[g]; p.sz[g] = p.az[g];
            }
            __syncthreads();
            permuted = true;
            pa.splits++;
            pa.iters += (unsigned long long)iters;
            emit_node(p, lane, frame, patch, depth, s, n, RPW_NODE_SPLIT, iters, count, axis, median, &pl);
            // children inherit mean_dist; process the smaller one first (bounded stack)
            const uint32_t nright = n - nleft;
            if (lane == 0) {
                const bool left_small = nleft <= nright;
                const uint32_t big_s = left_small ? s + nleft : s, big_n = left_small ? nright : nleft;
                const uint32_t sm_s = left_small ? s : s + nleft, sm_n = left_small ? nleft : nright;
                stk[3 * sp] = big_s; stk[3 * sp + 1] = big_n; stk[3 * sp + 2] = (uint32_t)(depth + 1);
                stk[3 * (sp + 1)] = sm_s; stk[3 * (sp + 1) + 1] = sm_n; stk[3 * (sp + 1) + 2] = (uint32_t)(depth + 1);
            }
            sp += 2;
            __syncthreads();
            continue;
        }

        // ---- FIT leaf: labels (and residuals) -----------------------------------------------------
        pa.iters += (unsigned long long)iters;
        for (uint32_t i = lane; i < n; i += kWave) {
            const uint32_t g = cx.base + s + i;
            const uint32_t orig = p.sidx[g];
            if ((p.mask[g] >> final_bit) & 1u) p.labels[fo + orig] = RPW_LABEL_GROUND;
            if (p.residual) {
                float x, y, z;
                if (permuted) { const P3 q = p.xyz[fo + orig]; x = q.x; y = q.y; z = q.z; }
                else { x = p.sx[g]; y = p.sy[g]; z = p.sz[g]; }
                p.residual[fo + orig] = plane_dist(x, y, z, pl.cx, pl.cy, pl.cz, pl.nx, pl.ny, pl.nz);
            }
        }
        pa.ground += count;
        emit_node(p, lane, frame, patch, depth, s, n, RPW_NODE_FIT, iters, count, -1, 0.0f, &pl);
    }

    if (lane == 0) {
        rpw_frame_stats* st = p.stats + frame;
        atomicAdd(&st->n_ground, pa.ground);
        atomicAdd(&st->n_nodes, pa.nodes);
        atomicAdd(&st->n_splits, pa.splits);
        atomicMax(&st->max_depth, pa.maxdepth);
        atomicAdd((unsigned long long*)&st->total_iterations, pa.iters);
    }
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
// launchers
// ------------------------------------------------------------------------------------------
void launch_classify(const DevParams& p, hipStream_t s)
{
    if (p.n_tiles == 0) return;
    hipLaunchKernelGGL(classify_hist_kernel, dim3(p.n_tiles), dim3(kTileThreads), (size_t)p.PB * 4, s, p);
}
void launch_scan(const DevParams& p, hipStream_t s)
{
    if (p.n_frames == 0) return;
    hipLaunchKernelGGL(scan_offsets_kernel, dim3(p.n_frames), dim3(kTileThreads), 0, s, p);
}
void launch_scatter(const DevParams& p, hipStream_t s)
{
    if (p.n_tiles == 0) return;
    hipLaunchKernelGGL(stable_scatter_kernel, dim3(p.n_tiles), dim3(kTileThreads), (size_t)p.PB * 5 * 4, s, p);
}
void launch_fit(const DevParams& p, hipStream_t s)
{
    const uint32_t grid = p.n_frames * (uint32_t)p.P;
    if (grid == 0) return;
    hipLaunchKernelGGL(patch_fit_kernel, dim3(grid), dim3(kWave), 0, s, p);
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
