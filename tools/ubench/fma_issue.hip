// Microbenchmark: per-wave issue rate of v_fma_f32 / v_fmac_f32 (SGPR operand)
// versus v_pk_fma_f32 (SGPR-pair operand, op_sel broadcast) at 1/2/4 waves per SIMD.
// Decides whether packed FMAs let 2 waves/SIMD saturate the fp32 VALU on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *taps, int iters)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    const float __attribute__((address_space(4))) *h = (const float __attribute__((address_space(4))) *)taps;
    float t0 = h[0], t1 = h[1];
    f2 tp = { t0, t1 };
    float x = threadIdx.x * 1e-9f;
    f2 xp = { x, x + 1.0f };
    if (MODE == 0) {
        float a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(t0), "v"(x));
        }
        float s = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += a[i];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f2 a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = f2{ (float)i, (float)-i };
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 1)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "s"(tp), "v"(xp));
                else
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(tp), "v"(xp));
            }
        }
        f2 s = { 0, 0 };
#pragma unroll
        for (int i = 0; i < 8; ++i) s += a[i];
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
    }
}

int main()
{
    float *out, *taps;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * 256 * 16));
    CHECK(hipMalloc(&taps, 64));
    float ht[2] = { 1e-3f, 2e-3f };
    CHECK(hipMemcpy(taps, ht, 8, hipMemcpyHostToDevice));
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[3] = { "v_fmac_f32 sgpr ", "v_pk_fma sgprpair", "v_pk_fma vgpr    " };
    for (int mode = 0; mode < 3; ++mode)
        for (int bpc = 1; bpc <= 8; bpc *= 2) {            // blocks of 4 waves per CU -> waves per SIMD
            dim3 g(256 * bpc), b(256);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, taps, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, taps, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, taps, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fma = (double)256 * bpc * 256 * 16.0 * iters;    // lane-FMAs
            printf("%s waves/SIMD=%d  %.3f ms  %.2f TFMA/s  (%.1f TFLOP/s)\n", names[mode], bpc, ms, fma / ms / 1e9, 2 * fma / ms / 1e9);
        }
    return 0;
}
