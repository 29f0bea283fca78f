// microbench: what does an instruction cost a lone wave when it sits (a) behind a group of four dependent v_add_f32,
// (b) between the dependent adds? Decides how the chain loops of patch_fit1 place the work that feeds the next slot.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o tools/microbench_fill tools/microbench_fill.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

#define FILL1(F) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(F) : "v"(c));
#define SB __builtin_amdgcn_sched_barrier(0);

// V: 0 = chain only; 1 = K fillers clumped behind each group of 4 adds; 2 = the same K fillers spread between the 4 adds
template <int V, int K>
__global__ void chain_fill(const float* in, float* out, long long* cyc, int n, int reps)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float* row = lds + (lane < 6 ? lane : 5) * 4;
    float acc = 0.f, f0 = 1.0f, f1 = 1.0f, f2 = 1.0f, f3 = 1.0f;
    const float c = 1.0000001f;
    typedef float v4f __attribute__((ext_vector_type(4)));
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        for (int j = 0; j + 64 <= n - 32; j += 64) {
            v4f A[8], B[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) A[q] = *reinterpret_cast<const v4f*>(row + j + 4 * q);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                B[q] = *reinterpret_cast<const v4f*>(row + j + 32 + 4 * q);
                if (V == 2) {
                    acc += A[q].x; SB if (K >= 1) { FILL1(f0) } if (K >= 5) { FILL1(f0) } SB
                    acc += A[q].y; SB if (K >= 2) { FILL1(f1) } if (K >= 6) { FILL1(f1) } SB
                    acc += A[q].z; SB if (K >= 3) { FILL1(f2) } if (K >= 7) { FILL1(f2) } SB
                    acc += A[q].w; SB if (K >= 4) { FILL1(f3) } if (K >= 8) { FILL1(f3) } SB
                } else {
                    acc += A[q].x; acc += A[q].y; acc += A[q].z; acc += A[q].w; SB
                    if (V == 1) {
                        if (K >= 1) { FILL1(f0) } if (K >= 2) { FILL1(f1) } if (K >= 3) { FILL1(f2) } if (K >= 4) { FILL1(f3) }
                        if (K >= 5) { FILL1(f0) } if (K >= 6) { FILL1(f1) } if (K >= 7) { FILL1(f2) } if (K >= 8) { FILL1(f3) }
                        SB
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { acc += B[q].x; acc += B[q].y; acc += B[q].z; acc += B[q].w; }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + f0 + f1 + f2 + f3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V, int K>
int run(const char* name, float* din, float* dout, long long* dc, int n)
{
    const int reps = 50;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    chain_fill<V, K><<<1, 64, n * 4>>>(din, dout, dc, n, reps);
    hipEventRecord(a); chain_fill<V, K><<<1, 64, n * 4>>>(din, dout, dc, n, reps); hipEventRecord(b);
    CK(hipDeviceSynchronize()); float ms; hipEventElapsedTime(&ms, a, b);
    // per 64 addends: 32 adds carry fillers (the A half), 32 do not (the B half)
    const double groups = (double)reps * ((n - 32) / 64);
    printf("%-44s %8.1f ns per 64 addends  (%.2f ns/add overall)\n", name, ms * 1e6 / groups, ms * 1e6 / groups / 64);
    return 0;
}

int main()
{
    const int n = 8192; std::vector<float> h(n, 1.0f);
    float *din, *dout; long long* dc;
    CK(hipMalloc(&din, n * 4)); CK(hipMalloc(&dout, 1 << 20)); CK(hipMalloc(&dc, 1024));
    CK(hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice));
    run<0, 0>("chain only", din, dout, dc, n);
    run<1, 1>("1 filler behind each 4 adds (x8 groups)", din, dout, dc, n);
    run<1, 2>("2 fillers behind each 4 adds", din, dout, dc, n);
    run<1, 4>("4 fillers behind each 4 adds", din, dout, dc, n);
    run<1, 8>("8 fillers behind each 4 adds", din, dout, dc, n);
    run<2, 2>("2 fillers, one between adds each", din, dout, dc, n);
    run<2, 4>("4 fillers, one after every add", din, dout, dc, n);
    run<2, 8>("8 fillers, two after every add", din, dout, dc, n);
    return 0;
}
