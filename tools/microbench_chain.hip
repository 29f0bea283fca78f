// microbench: cost of a sequential binary32 add chain fed from LDS (the K4 "chain lane" loop)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

__global__ void chain_lds(const float* in, float* out, long long* cyc, int n, int reps)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float* row = lds + (lane < 6 ? lane : 5) * 4;   // tiny offset per lane, same structure as K4
    float acc = 0.f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        for (int j = 0; j + 64 <= n - 32; j += 64) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { float4 v = *reinterpret_cast<const float4*>(row + j + 4 * q); acc += v.x; acc += v.y; acc += v.z; acc += v.w; }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void chain_reg(float* out, long long* cyc, int iters, float a)
{
    float acc = 0.f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < iters; ++r) {
#pragma unroll
        for (int q = 0; q < 64; ++q) { acc += a; asm volatile("" : "+v"(acc)); }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    const int n = 8192; std::vector<float> h(n, 1.0f);
    float *din, *dout; long long* dc;
    CK(hipMalloc(&din, n * 4)); CK(hipMalloc(&dout, 1 << 22)); CK(hipMalloc(&dc, 8192 * 8));
    CK(hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<long long> hc(8192);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("clock rate attr %d kHz\n", clk);
    for (int threads : {64, 128, 256, 512}) for (int blocks : {1, 256, 1024, 2048}) {
        const int reps = 50;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        chain_lds<<<blocks, threads, n * 4>>>(din, dout, dc, n, reps);
        hipEventRecord(a); chain_lds<<<blocks, threads, n * 4>>>(din, dout, dc, n, reps); hipEventRecord(b);
        CK(hipDeviceSynchronize()); float ms; hipEventElapsedTime(&ms, a, b);
        CK(hipMemcpy(hc.data(), dc, blocks * 8, hipMemcpyDeviceToHost));
        const double adds = (double)reps * ((n - 32) / 64) * 64;
        printf("lds chain: threads %4d blocks %5d  memtime ticks/add %.2f  wall ns/add %.3f\n", threads, blocks, hc[0] / adds, ms * 1e6 / adds);
    }
    for (int threads : {64, 256}) for (int blocks : {1, 1024}) {
        const int iters = 20000; hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        chain_reg<<<blocks, threads>>>(dout, dc, iters, 1.0f);
        hipEventRecord(a); chain_reg<<<blocks, threads>>>(dout, dc, iters, 1.0f); hipEventRecord(b);
        CK(hipDeviceSynchronize()); float ms; hipEventElapsedTime(&ms, a, b);
        CK(hipMemcpy(hc.data(), dc, blocks * 8, hipMemcpyDeviceToHost));
        printf("reg chain: threads %4d blocks %5d  memtime ticks/add %.2f  wall ns/add %.3f\n", threads, blocks, hc[0] / (iters * 64.0), ms * 1e6 / (iters * 64.0));
    }
    return 0;
}
