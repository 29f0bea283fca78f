// microbench: the per-slot loop of the one-feeder chain server (rpw_kernels.hip srv1_loop), piece by piece: which of its
// parts makes a slot of 64 dependent adds cost more than 64 x 8.2 cycles?
//   bit 0: the published count is read from LDS and decides (v_readfirstlane + scalar branch) whether the next slot is prefetched
//   bit 1: the accumulator and the slot count are written to LDS after every slot (exec-masked stores)
//   bit 2: the "first slot of a pass" test in front of every slot
//   bit 3: sched_barrier between the groups of four adds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) v4f lds_f4;
typedef __attribute__((address_space(3))) uint32_t lds_word;
constexpr int kLdw = 68, kSlotFloats = 6 * kLdw;
#define CHAIN4(ACC, Q) { ACC += (Q).x; ACC += (Q).y; ACC += (Q).z; ACC += (Q).w; }

template <int F>
__global__ __launch_bounds__(64) void srv1_like(float* out, long long* cyc, uint32_t nslots, int reps)
{
    __shared__ __attribute__((aligned(16))) float ring[2048];
    __shared__ uint32_t words[8];
    __shared__ float accs[8];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) ring[i] = 1.0f + i * 1.0e-3f;
    if (lane == 0) { words[0] = nslots; words[1] = 0; }
    __syncthreads();
    const lds_f32* ringrow = (lds_f32*)ring + (lane < 6 ? lane : 5) * kLdw;
    volatile lds_word* pubw = (volatile lds_word*)&words[0];
    lds_f32* accw = (lds_f32*)&accs[lane < 6 ? lane : 5];
    lds_word* consw = (lds_word*)&words[2];
    float acc = 0.0f;
    v4f A[8], B[8];
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        uint32_t cons = 0; const uint32_t start = 0;
        const lds_f32* src = ringrow + ((cons - start) & 3u) * kSlotFloats;
#pragma unroll
        for (int q = 0; q < 8; ++q) A[q] = *reinterpret_cast<const lds_f4*>(src + 4 * q);
        for (;;) {
            if (F & 4) { if (cons == start) acc = 0.0f; }
            uint32_t pubv = nslots;
            if (F & 1) pubv = *pubw;
            if (F & 8) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; ++q) { B[q] = *reinterpret_cast<const lds_f4*>(src + 32 + 4 * q); CHAIN4(acc, A[q]) if (F & 8) __builtin_amdgcn_sched_barrier(0); }
            ++cons;
            const bool more = (int32_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)pubv) - cons) > 0;
            if (more) {
                src = ringrow + ((cons - start) & 3u) * kSlotFloats;
                if (F & 8) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; ++q) { A[q] = *reinterpret_cast<const lds_f4*>(src + 4 * q); CHAIN4(acc, B[q]) if (F & 8) __builtin_amdgcn_sched_barrier(0); }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) { CHAIN4(acc, B[q]) }
            }
            if (F & 2) {
                asm volatile("" ::: "memory");
                if (lane < 6) *(volatile lds_f32*)accw = acc;
                if (lane == 0) *(volatile lds_word*)consw = cons;
            }
            if (!more) break;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int F> int run(float* dout, long long* dc)
{
    const uint32_t nslots = 128; const int reps = 50;
    srv1_like<F><<<1, 64>>>(dout, dc, nslots, reps);
    srv1_like<F><<<1, 64>>>(dout, dc, nslots, reps);
    CK(hipDeviceSynchronize());
    long long c = 0; CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));
    printf("flags %2d: %.2f cycles per add (%.0f per slot)\n", F, (double)c / (64.0 * nslots * reps), (double)c / (nslots * reps));
    return 0;
}

int main()
{
    float* dout; long long* dc;
    CK(hipMalloc(&dout, 4096)); CK(hipMalloc(&dc, 64));
    run<0>(dout, dc); run<8>(dout, dc); run<1>(dout, dc); run<9>(dout, dc); run<2>(dout, dc); run<10>(dout, dc); run<4>(dout, dc); run<12>(dout, dc);
    run<3>(dout, dc); run<11>(dout, dc); run<15>(dout, dc); run<7>(dout, dc);
    return 0;
}
