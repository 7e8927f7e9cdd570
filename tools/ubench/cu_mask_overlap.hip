// Microbenchmark (round 6, the 1 M / 1.6 M / 2 MS/s plans): can a cascade's SECOND stage run beside the NEXT batch's first
// stage on a handful of CUs of its own?  A k_fir_i8x block fills its CU (768 threads, 139 KB of LDS), so a second kernel
// only ever runs where no such block sits: two streams with complementary CU masks (hipExtStreamCreateWithCUMask).
//   A  the first stage's traffic: persistent blocks of 768 threads, one per CU of ITS mask, tiles of 10240 samples (61 KB,
//      three 16-byte loads per lane and group, two tiles in flight), 8 KB of nontemporal stores per tile -- 2^28 samples
//   B  the second stage's traffic: 2^28 / 10 float2 in (215 MB), a quarter of that out (54 MB), 256-thread blocks, on the
//      other CUs
// Times of A alone on all CUs, A on its share, B on its share, and both at once (events on each stream).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(768, 1) void kA(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, int ntiles)
{
    extern __shared__ unsigned char lds[];                 // claimed so that nothing else shares the CU (139 KB, like the product)
    const int tid = threadIdx.x;
    u32x4 acc = { 0, 0, 0, 0 };
    // tile t: 1280 groups of 48 bytes; loader-like: threads 256..767 load, two and a half rounds
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        if (tid >= 256) {
            const int lt = tid - 256;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int g = lt + 512 * q;
                if (g < 1280) {
                    const u32x4 *p = in + ((size_t)t * 1280 + g) * 3;
                    acc ^= p[0];
                    acc ^= p[1];
                    acc ^= p[2];
                }
            }
        }
        if (tid < 256) {                                   // 1024 outputs x 8 B = 8 KB = 512 x 16 B
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 *d = out + (size_t)t * 512 + tid + 256 * h;
                const u32x4 v = { (unsigned)t, (unsigned)tid, 1u, 2u };
                asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(d), "v"(v) : "memory");
            }
        }
    }
    if (acc.x == 0x12345678u && lds[tid] == 77)
        out[0] = acc;
}

__global__ __launch_bounds__(256) void kB(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n16_in)
{
    // every block streams a contiguous share; four 16-byte loads per thread and step, one 16-byte store per four loads
    const size_t per = (n16_in / gridDim.x) & ~(size_t)1023, b0 = per * blockIdx.x;
    u32x4 acc = { 0, 0, 0, 0 };
    for (size_t i = b0 + threadIdx.x; i + 768 < b0 + per; i += 1024) {
        const u32x4 a = in[i], b = in[i + 256], c = in[i + 512], d = in[i + 768];
        acc = a ^ b ^ c ^ d;
        u32x4 *dst = out + (i >> 2);
        asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(dst), "v"(acc) : "memory");
    }
}



#include <functional>
static float timed2(hipStream_t sa, hipStream_t sb, int reps, const std::function<void()> &fa, const std::function<void()> &fb, float *tb)
{
    hipEvent_t a0, a1, b0, b1;
    hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
    for (int i = 0; i < 5; ++i) { fa(); if (fb) fb(); }
    hipDeviceSynchronize();
    hipEventRecord(a0, sa);
    if (fb) hipEventRecord(b0, sb);
    for (int i = 0; i < reps; ++i) { fa(); if (fb) fb(); }
    hipEventRecord(a1, sa);
    if (fb) hipEventRecord(b1, sb);
    hipDeviceSynchronize();
    float ta = 0, t2 = 0;
    hipEventElapsedTime(&ta, a0, a1);
    if (fb) hipEventElapsedTime(&t2, b0, b1);
    if (tb) *tb = t2 / reps;
    return ta / reps;
}

int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const size_t ns = (size_t)1 << 28;
    const int ntiles = (int)(ns / 10240);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    char *in, *outA, *mid, *outB;
    CHECK(hipMalloc(&in, ns * 6 + 4096));
    CHECK(hipMalloc(&outA, ns / 10 * 8 + 4096));
    CHECK(hipMalloc(&mid, ns / 10 * 8 + 65536));
    CHECK(hipMalloc(&outB, ns / 40 * 8 + 65536));
    CHECK(hipMemset(in, 1, ns * 6));
    CHECK(hipMemset(mid, 1, ns / 10 * 8));
    const size_t ldsA = 139 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&kA), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsA));
    printf("%d CUs.  A = first-stage traffic (2^28 samples), B = second-stage traffic (215 MB in, 54 MB out)\n", ncu);
    hipStream_t s_all;
    CHECK(hipStreamCreate(&s_all));
    auto runA = [&](hipStream_t s, int blocks) { hipLaunchKernelGGL(kA, dim3(blocks), dim3(768), ldsA, s, (const u32x4 *)in, (u32x4 *)outA, ntiles); };
    auto runB = [&](hipStream_t s, int blocks) { hipLaunchKernelGGL(kB, dim3(blocks), dim3(256), 0, s, (const u32x4 *)mid, (u32x4 *)outB, ns / 10 * 8 / 16); };
    const float a_all = timed2(s_all, s_all, 20, [&] { runA(s_all, ncu); }, nullptr, nullptr);
    const float b_all = timed2(s_all, s_all, 20, [&] { runB(s_all, ncu * 8); }, nullptr, nullptr);
    printf("unmasked: A on %d CUs %.4f ms; B on all CUs %.4f ms; one after the other %.4f ms\n", ncu, a_all, b_all, a_all + b_all);
    for (int nb : { 8, 16, 24, 32, 48 }) {
        // B's CUs: spread evenly over the mask's bit positions (bit i = CU i in the runtime's numbering, XCDs interleaved)
        std::vector<uint32_t> mA((ncu + 31) / 32, 0), mB((ncu + 31) / 32, 0);
        for (int i = 0; i < ncu; ++i) {
            const bool forB = (i % (ncu / nb)) == 0 && (i / (ncu / nb)) < nb;
            (forB ? mB : mA)[i / 32] |= 1u << (i % 32);
        }
        hipStream_t sa, sb;
        printf("masks for %d CUs of B ...\n", nb);
        CHECK(hipExtStreamCreateWithCUMask(&sa, (uint32_t)mA.size(), mA.data()));
        CHECK(hipExtStreamCreateWithCUMask(&sb, (uint32_t)mB.size(), mB.data()));
        printf("streams made\n");
        const float a_share = timed2(sa, sa, 20, [&] { runA(sa, ncu - nb); }, nullptr, nullptr);
        const float b_share = timed2(sb, sb, 20, [&] { runB(sb, nb * 8); }, nullptr, nullptr);
        float b_both = 0;
        const float a_both = timed2(sa, sb, 20, [&] { runA(sa, ncu - nb); }, [&] { runB(sb, nb * 8); }, &b_both);
        printf("B on %2d CUs: A alone on %3d CUs %.4f ms | B alone %.4f ms | together: A %.4f ms, B %.4f ms -> a step of %.4f ms (now %.4f)\n", nb,
               ncu - nb, a_share, b_share, a_both, b_both, a_both > b_both ? a_both : b_both, a_all + b_all);
        hipStreamDestroy(sa);
        hipStreamDestroy(sb);
    }
    return 0;
}
