// Microbenchmark: the unpack-only traffic (6 B in, 8 B out per sample) without arithmetic:
// what k_unpack24 (14 B/sample) can reach at best.  Also a plain 1:1 copy of the same volume.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// lane-contiguous 16-byte loads and stores (the best-coalesced shape): 3 loads + 4 stores per 8 samples
template <int NT>
__global__ __launch_bounds__(256) void k(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t ngroups)
{
    const size_t nth = (size_t)gridDim.x * 256;
    for (size_t g0 = (size_t)blockIdx.x * 256; g0 < ngroups; g0 += nth) {
        // a block handles 256 groups: 768 input chunks, 1024 output chunks, all lane-contiguous
        const u32x4 *src = in + g0 * 3;
        u32x4 *dst = out + g0 * 4;
        u32x4 v[3];
#pragma unroll
        for (int k2 = 0; k2 < 3; ++k2) v[k2] = src[threadIdx.x + 256 * k2];
        const u32x4 w = { v[0].x ^ v[1].y, v[1].z + v[2].w, v[2].x ^ v[0].w, v[0].z + v[1].x };
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            if (NT) __builtin_nontemporal_store(k2 < 3 ? v[k2] : w, dst + threadIdx.x + 256 * k2);
            else dst[threadIdx.x + 256 * k2] = k2 < 3 ? v[k2] : w;
        }
    }
}

int main()
{
    const size_t ns = (size_t)1 << 28;
    u32x4 *in, *out;
    CHECK(hipMalloc(&in, ns * 6));
    CHECK(hipMalloc(&out, ns * 8));
    CHECK(hipMemset(in, 1, ns * 6));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nt = 0; nt < 2; ++nt)
        for (int blocks : { 512, 1024, 2048, 8192 }) {
            float best = 1e9;
            for (int rep = 0; rep < 8; ++rep) {
                hipEventRecord(e0);
                if (nt) hipLaunchKernelGGL((k<1>), dim3(blocks), dim3(256), 0, 0, in, out, ns / 8);
                else hipLaunchKernelGGL((k<0>), dim3(blocks), dim3(256), 0, 0, in, out, ns / 8);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("6 B in + 8 B out per sample, %s stores, %5d blocks: %.3f ms = %.2f TB/s\n", nt ? "nt   " : "plain", blocks, best,
                   ns * 14.0 / best / 1e9);
        }
    return 0;
}
