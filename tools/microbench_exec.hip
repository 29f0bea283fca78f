// microbench: does a dependent v_add_f32 chain (or a saturated SIMD) get cheaper when only the first L lanes are enabled?
// (patch_fit1's chains use 3 or 6 lanes of 64.)   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o tools/microbench_exec tools/microbench_exec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
template <int L>
__global__ void chain(const float* in, float* out, int reps)
{
    const int lane = threadIdx.x & 63;
    float acc = in[lane], a = in[64 + lane];
    if (lane < L) {
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int k = 0; k < 64; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(a));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int L>
int run(float* din, float* dout, int blocks, int threads)
{
    const int reps = 2000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    chain<L><<<blocks, threads>>>(din, dout, reps);
    CK(hipEventRecord(a)); chain<L><<<blocks, threads>>>(din, dout, reps); CK(hipEventRecord(b));
    CK(hipDeviceSynchronize()); float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double adds = (double)reps * 64;
    printf("lanes %2d  blocks %5d x %4d threads: %7.3f ms  %.2f ns per add per wave (%.2f cycles @2.4GHz)\n", L, blocks, threads, ms, ms * 1e6 / adds, ms * 1e6 / adds * 2.4);
    return 0;
}
int main()
{
    float *din, *dout; CK(hipMalloc(&din, 4096)); CK(hipMalloc(&dout, 64 << 20)); CK(hipMemset(din, 0, 4096));
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int blocks = cfg == 0 ? 1 : 1024 * (cfg == 1 ? 1 : 2), threads = cfg == 0 ? 64 : 256 * (cfg == 1 ? 1 : 2);
        // cfg 1: one wave per SIMD on the whole chip; cfg 2: 8 waves per SIMD (1024 blocks... of 512 threads x2)
        run<64>(din, dout, blocks, threads); run<32>(din, dout, blocks, threads); run<16>(din, dout, blocks, threads); run<3>(din, dout, blocks, threads);
    }
    return 0;
}
