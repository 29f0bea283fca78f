// microbench: shapes of "64 dependent adds fed by 16 ds_read_b128" -- how should the LDS reads of the next addends be placed
// among the adds of a lone chain wave?
//   shape 0: one read between every group of four adds (the sweeps' shape), halves A / B
//   shape 1: the eight reads of the next half all at once, then 32 adds
//   shape 2: the sixteen reads of the next slot all at once, then 64 adds
//   shape 3: like 0, but the reads as ds_read_b64 pairs (two per group)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) v4f lds_f4;
typedef __attribute__((address_space(3))) v2f lds_f2;
constexpr int kLdw = 68, kSlotFloats = 6 * kLdw;
#define CHAIN4(ACC, Q) { ACC += (Q).x; ACC += (Q).y; ACC += (Q).z; ACC += (Q).w; }

template <int SHAPE>
__global__ __launch_bounds__(64) void shape(float* out, long long* cyc, uint32_t nslots, int reps)
{
    __shared__ __attribute__((aligned(16))) float ring[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) ring[i] = 1.0f + i * 1.0e-3f;
    __syncthreads();
    const lds_f32* ringrow = (lds_f32*)ring + (lane < 6 ? lane : 5) * kLdw;
    float acc = 0.0f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if (SHAPE == 0 || SHAPE == 3) {
            v4f A[8], B[8];
            const lds_f32* src = ringrow;
#pragma unroll
            for (int q = 0; q < 8; ++q) A[q] = *reinterpret_cast<const lds_f4*>(src + 4 * q);
            for (uint32_t s = 0; s < nslots; ++s) {
                const lds_f32* nxt = ringrow + ((s + 1) & 3u) * kSlotFloats;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (SHAPE == 0) B[q] = *reinterpret_cast<const lds_f4*>(src + 32 + 4 * q);
                    else { const v2f a = *reinterpret_cast<const lds_f2*>(src + 32 + 4 * q), b = *reinterpret_cast<const lds_f2*>(src + 34 + 4 * q); B[q] = v4f{a.x, a.y, b.x, b.y}; }
                    CHAIN4(acc, A[q])
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (SHAPE == 0) A[q] = *reinterpret_cast<const lds_f4*>(nxt + 4 * q);
                    else { const v2f a = *reinterpret_cast<const lds_f2*>(nxt + 4 * q), b = *reinterpret_cast<const lds_f2*>(nxt + 2 + 4 * q); A[q] = v4f{a.x, a.y, b.x, b.y}; }
                    CHAIN4(acc, B[q])
                }
                src = nxt;
            }
        } else if (SHAPE == 1) {
            v4f A[8], B[8];
            const lds_f32* src = ringrow;
#pragma unroll
            for (int q = 0; q < 8; ++q) A[q] = *reinterpret_cast<const lds_f4*>(src + 4 * q);
            for (uint32_t s = 0; s < nslots; ++s) {
                const lds_f32* nxt = ringrow + ((s + 1) & 3u) * kSlotFloats;
#pragma unroll
                for (int q = 0; q < 8; ++q) B[q] = *reinterpret_cast<const lds_f4*>(src + 32 + 4 * q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; ++q) CHAIN4(acc, A[q])
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; ++q) A[q] = *reinterpret_cast<const lds_f4*>(nxt + 4 * q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; ++q) CHAIN4(acc, B[q])
                __builtin_amdgcn_sched_barrier(0);
                src = nxt;
            }
        } else if (SHAPE >= 4) {
            // shape 1 + what the server needs besides: bit 0 (5): the published count read in the middle and tested at the end,
            // bit 1 (6): two unmasked LDS stores per slot, 7: both
            __shared__ uint32_t words[4]; __shared__ float accs[64]; __shared__ uint32_t dumpw[64];
            if (lane == 0) words[0] = nslots + 1;
            __syncthreads();
            volatile __attribute__((address_space(3))) uint32_t* pubw = (volatile __attribute__((address_space(3))) uint32_t*)&words[0];
            volatile lds_f32* accw = (volatile lds_f32*)&accs[lane];
            volatile __attribute__((address_space(3))) uint32_t* consw = (volatile __attribute__((address_space(3))) uint32_t*)&dumpw[lane];
            v4f A[8], B[8];
            const lds_f32* src = ringrow;
            uint32_t cons = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) A[q] = *reinterpret_cast<const lds_f4*>(src + 4 * q);
            for (;;) {
#pragma unroll
                for (int q = 0; q < 8; ++q) B[q] = *reinterpret_cast<const lds_f4*>(src + 32 + 4 * q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; ++q) CHAIN4(acc, A[q])
                __builtin_amdgcn_sched_barrier(0);
                ++cons;
                uint32_t pubv = nslots + 1;
                if ((SHAPE - 4) & 1) pubv = *pubw;
                src = ringrow + (cons & 3u) * kSlotFloats;
#pragma unroll
                for (int q = 0; q < 8; ++q) A[q] = *reinterpret_cast<const lds_f4*>(src + 4 * q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; ++q) CHAIN4(acc, B[q])
                __builtin_amdgcn_sched_barrier(0);
                if ((SHAPE - 4) & 2) { *accw = acc; *consw = cons; }
                if ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)pubv) - cons) <= 0 || cons >= nslots) break;
            }
        } else {
            v4f A[16], N[16];
            const lds_f32* src = ringrow;
#pragma unroll
            for (int q = 0; q < 16; ++q) A[q] = *reinterpret_cast<const lds_f4*>(src + 4 * q);
            for (uint32_t s = 0; s < nslots; s += 2) {
                const lds_f32* n1 = ringrow + ((s + 1) & 3u) * kSlotFloats;
                const lds_f32* n2 = ringrow + ((s + 2) & 3u) * kSlotFloats;
#pragma unroll
                for (int q = 0; q < 16; ++q) N[q] = *reinterpret_cast<const lds_f4*>(n1 + 4 * q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 16; ++q) CHAIN4(acc, A[q])
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 16; ++q) A[q] = *reinterpret_cast<const lds_f4*>(n2 + 4 * q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 16; ++q) CHAIN4(acc, N[q])
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int S> int run(float* dout, long long* dc)
{
    const uint32_t nslots = 128; const int reps = 50;
    shape<S><<<1, 64>>>(dout, dc, nslots, reps);
    shape<S><<<1, 64>>>(dout, dc, nslots, reps);
    CK(hipDeviceSynchronize());
    long long c = 0; CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));
    printf("shape %d: %.2f cycles per add (%.0f per slot)\n", S, (double)c / (64.0 * nslots * reps), (double)c / (nslots * reps));
    return 0;
}

int main()
{
    float* dout; long long* dc;
    CK(hipMalloc(&dout, 4096)); CK(hipMalloc(&dc, 64));
    run<0>(dout, dc); run<1>(dout, dc); run<2>(dout, dc); run<3>(dout, dc); run<4>(dout, dc); run<5>(dout, dc); run<6>(dout, dc); run<7>(dout, dc);
    return 0;
}
