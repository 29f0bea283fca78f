// Where do the waves of a small workgroup run? Prints the SIMD of every wave of a few workgroups of 2, 4 and 8 waves
// (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8]). If the two waves of patch_fit_srv1_kernel's workgroup share
// a SIMD, the feeder's vector instructions delay the server's dependent adds.
//   hipcc -O3 --offload-arch=gfx950 -o tools/microbench_placement tools/microbench_placement.hip && tools/microbench_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void where(unsigned* out)
{
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
}
__global__ void which_xcd(unsigned* out)
{
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) out[blockIdx.x] = id & 15u;
}
int main()
{
    {   // the XCD of consecutive workgroups of a large grid (256 threads each, like the binning tiles)
        const int blocks = 4096;
        unsigned* d; hipMalloc(&d, blocks * 4);
        hipLaunchKernelGGL(which_xcd, dim3(blocks), dim3(256), 0, 0, d);
        std::vector<unsigned> h(blocks);
        hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost);
        printf("XCD of workgroups 0..31:");
        for (int b = 0; b < 32; ++b) printf(" %u", h[b]);
        int rr = 0; unsigned cnt[16] = {0};
        for (int b = 0; b < blocks; ++b) { rr += h[b] == (unsigned)((h[0] + b) % 8); cnt[h[b]]++; }
        printf("\nworkgroups whose XCD is (XCD of workgroup 0 + index) mod 8: %d of %d; per XCD:", rr, blocks);
        for (int x = 0; x < 8; ++x) printf(" %u", cnt[x]);
        printf("\n");
        hipFree(d);
    }
    for (int waves : {2, 4, 8}) {
        const int blocks = 6;
        unsigned* d; hipMalloc(&d, blocks * waves * 4);
        hipLaunchKernelGGL(where, dim3(blocks), dim3(64 * waves), 0, 0, d);
        std::vector<unsigned> h(blocks * waves);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        printf("%d waves per workgroup: (cu.simd) per wave\n", waves);
        for (int b = 0; b < blocks; ++b) {
            printf("  wg %d:", b);
            for (int w = 0; w < waves; ++w) { const unsigned v = h[b * waves + w]; printf(" %u.%u", (v >> 8) & 15u, (v >> 4) & 3u); }
            printf("\n");
        }
        hipFree(d);
    }
    return 0;
}
