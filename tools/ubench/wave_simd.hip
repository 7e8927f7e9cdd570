// Which SIMD does wave w of a 768-thread workgroup run on?  (HW_REG_HW_ID: wave_id[3:0], simd_id[5:4], ..., cu_id[11:8])
// build: hipcc --offload-arch=gfx950 -O2 -o wave_simd wave_simd.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(768, 1) void k(unsigned *out)
{
    extern __shared__ unsigned char lds[];
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0)
        out[blockIdx.x * 12 + (threadIdx.x >> 6)] = hw;
    if (threadIdx.x == 100000) lds[0] = 1;
}
int main()
{
    unsigned *d, h[12 * 16];
    hipMalloc(&d, sizeof(h));
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(16), dim3(768), 150 * 1024, 0, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int b = 0; b < 16; b += 5) {
            printf("block %2d:", b);
            for (int w = 0; w < 12; ++w)
                printf(" w%d->simd%u(slot%u,cu%u)", w, (h[b * 12 + w] >> 4) & 3, h[b * 12 + w] & 15, (h[b * 12 + w] >> 8) & 15);
            printf("\n");
        }
    }
    return 0;
}
