// microbench: what does the shape of a wave's loads cost at full occupancy? patch_fit's passes read x, y, z of consecutive slots
// (fused pass) or of the ~56 % of the slots that are inliers (covariance sweep, through an index list). Compared here:
//   SoA  three arrays, one dword load each per lane                 (today's layout: sx, sy, sz)
//   AoS  one array of 12-byte records, one dwordx3 load per lane
// for a streaming read and for an ascending gather, on a working set that fits the Infinity Cache and on one that does not.
//   hipcc -O3 --offload-arch=gfx950 -o tools/microbench_loads tools/microbench_loads.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
struct P3 { float x, y, z; };

// every block walks its own contiguous region of `per_block` slots, `reps` times (like a patch's refits)
template <int AOS, int GATHER>
__global__ __launch_bounds__(64) void walk(const float* sx, const float* sy, const float* sz, const P3* sp, const unsigned short* il,
                                           unsigned per_block, unsigned count, int reps, float* out)
{
    const size_t base = (size_t)blockIdx.x * per_block;
    const unsigned lane = threadIdx.x;
    float acc = 0.f;
    const unsigned n = GATHER ? count : per_block;
    for (int r = 0; r < reps; ++r) {
        for (unsigned j0 = 0; j0 + 256 <= n; j0 += 256) {
            float a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                unsigned j = j0 + u * 64 + lane;
                const size_t i = base + (GATHER ? (unsigned)il[base + j] : j);
                if (AOS) { const P3 p = sp[i]; a[u] = p.x + p.y + p.z; }
                else a[u] = sx[i] + sy[i] + sz[i];
            }
            acc += (a[0] + a[1]) + (a[2] + a[3]);
        }
    }
    out[blockIdx.x * 64 + lane] = acc;
}

template <int AOS, int GATHER>
int run(const char* name, const float* sx, const float* sy, const float* sz, const P3* sp, const unsigned short* il, unsigned blocks, unsigned per_block, unsigned count, float* out)
{
    const int reps = 20;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    walk<AOS, GATHER><<<blocks, 64>>>(sx, sy, sz, sp, il, per_block, count, 1, out);
    CK(hipEventRecord(a)); walk<AOS, GATHER><<<blocks, 64>>>(sx, sy, sz, sp, il, per_block, count, reps, out); CK(hipEventRecord(b));
    CK(hipDeviceSynchronize()); float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double elems = (double)reps * blocks * (GATHER ? count : per_block);
    printf("  %-34s %8.3f ms  %7.1f G elements/s  %6.2f TB/s of x,y,z\n", name, ms, elems / ms / 1e6, elems * 12 / ms / 1e9);
    return 0;
}

int main()
{
    const unsigned blocks = 256 * 16;
    for (unsigned per_block : {2048u, 16384u}) {
        const size_t n = (size_t)blocks * per_block;
        float *sx, *sy, *sz, *out; P3* sp; unsigned short* il;
        CK(hipMalloc(&sx, n * 4)); CK(hipMalloc(&sy, n * 4)); CK(hipMalloc(&sz, n * 4)); CK(hipMalloc(&sp, n * 12)); CK(hipMalloc(&il, n * 2)); CK(hipMalloc(&out, blocks * 64 * 4));
        CK(hipMemset(sx, 0, n * 4)); CK(hipMemset(sy, 0, n * 4)); CK(hipMemset(sz, 0, n * 4)); CK(hipMemset(sp, 0, n * 12));
        // ascending index list with 56 % density, the same for every block
        std::vector<unsigned short> h(per_block); unsigned cnt = 0; unsigned seed = 1;
        for (unsigned i = 0; i < per_block; ++i) { seed = seed * 1664525u + 1013904223u; if ((seed >> 8) % 100 < 56) h[cnt++] = (unsigned short)i; }
        cnt &= ~255u;
        for (unsigned b = 0; b < blocks; ++b) CK(hipMemcpy(il + (size_t)b * per_block, h.data(), per_block * 2, hipMemcpyHostToDevice));
        printf("%u waves (16 per CU), %u slots each: %.0f MB of x,y,z\n", blocks, per_block, n * 12 / 1e6);
        run<0, 0>("stream, SoA 3 x dword", sx, sy, sz, sp, il, blocks, per_block, cnt, out);
        run<1, 0>("stream, AoS dwordx3", sx, sy, sz, sp, il, blocks, per_block, cnt, out);
        run<0, 1>("gather 56 %, SoA 3 x dword", sx, sy, sz, sp, il, blocks, per_block, cnt, out);
        run<1, 1>("gather 56 %, AoS dwordx3", sx, sy, sz, sp, il, blocks, per_block, cnt, out);
        hipFree(sx); hipFree(sy); hipFree(sz); hipFree(sp); hipFree(il); hipFree(out);
    }
    return 0;
}
