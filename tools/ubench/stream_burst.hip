// Microbenchmark: k_fir8's traffic (6 B/sample in, 1 B/sample out, persistent blocks with
// contiguous tile ranges, 48-B-per-lane loads, nt dwordx4 stores), no compute.  Does it
// matter whether a block writes its 8 KB of output after every tile or saves up B tiles and
// writes B*8 KB at once (fewer read/write turnarounds in the memory controllers)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int B, int FL>
__global__ __launch_bounds__(256) void k(const uint4 *__restrict__ in, uint4 *__restrict__ out, int ntiles, int tpb)
{
    // tile = 8192 samples = 1024 groups of 48 B (4 per thread); output 8 KB = 512 uint4 (2 per thread)
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * tpb, t1 = min(t0 + tpb, ntiles);
    uint4 keep[B][2];
    for (int tb = t0; tb < t1; tb += B) {
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int t = tb + b;
            uint4 acc = make_uint4(0, 0, 0, 0);
            if (t < t1) {
                const uint4 *src = in + ((size_t)t * 1024 + tid) * 3;
                uint4 v[4][3];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int w = 0; w < 3; ++w) v[k][w] = src[(size_t)k * 256 * 3 + w];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int w = 0; w < 3; ++w) { acc.x ^= v[k][w].x; acc.y += v[k][w].y; acc.z ^= v[k][w].z; acc.w += v[k][w].w; }
            }
            keep[b][0] = acc;
            keep[b][1] = make_uint4(acc.y, acc.x, acc.w, acc.z);
        }
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int t = tb + b;
            if (t < t1) {
                uint4 *dst = out + (size_t)t * 512;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4 v = { keep[b][h].x, keep[b][h].y, keep[b][h].z, keep[b][h].w };
                    uint4 *d = dst + tid + 256 * h;
                    if (FL == 0) __builtin_nontemporal_store(v, (u32x4 *)d);
                    else if (FL == 1) *(u32x4 *)d = v;
                    else if (FL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(d), "v"(v) : "memory");
                    else if (FL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(d), "v"(v) : "memory");
                    else if (FL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" : : "v"(d), "v"(v) : "memory");
                    else asm volatile("global_store_dwordx4 %0, %1, off sc0 nt\n\ts_nop 1" : : "v"(d), "v"(v) : "memory");
                }
            }
        }
    }
}

template <int B, int FL = 0>
static float run(const uint4 *in, uint4 *out, size_t ns, int blocks)
{
    const int ntiles = (int)(ns / 8192);
    int tpb = (ntiles + blocks - 1) / blocks;
    tpb = (tpb + B - 1) / B * B;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 8; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<B, FL>), dim3(blocks), dim3(256), 0, 0, in, out, ntiles, tpb);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    const size_t ns = (size_t)1 << 28;
    uint4 *in, *out;
    CHECK(hipMalloc(&in, ns * 6));
    // SPACER_GIB=n: n GiB of other allocations between the input and the output, so that the two land in
    // different HBM extent classes (NOTEBOOK.md rounds 1-3 5 (o)); the default, 0, is the first-come placement
    if (const char *e = getenv("SPACER_GIB")) {
        for (int k = 0; k < atoi(e); k += 8) {
            void *sp;
            CHECK(hipMalloc(&sp, (size_t)8 << 30));
        }
    }
    CHECK(hipMalloc(&out, ns));
    CHECK(hipMemset(in, 1, ns * 6));
    for (int blocks : { 512, 1024 })
        printf("blocks=%d: write every tile %.3f ms, every 2 tiles %.3f, every 4 tiles %.3f, every 8 tiles %.3f\n", blocks,
               run<1>(in, out, ns, blocks), run<2>(in, out, ns, blocks), run<4>(in, out, ns, blocks), run<8>(in, out, ns, blocks));
    printf("store flavour (512 blocks, write every tile / every 8): nt %.3f/%.3f  plain %.3f/%.3f  sc0sc1 %.3f/%.3f  sc1 %.3f/%.3f  sc0sc1nt %.3f/%.3f  sc0nt %.3f/%.3f\n",
           run<1, 0>(in, out, ns, 512), run<8, 0>(in, out, ns, 512), run<1, 1>(in, out, ns, 512), run<8, 1>(in, out, ns, 512),
           run<1, 2>(in, out, ns, 512), run<8, 2>(in, out, ns, 512), run<1, 3>(in, out, ns, 512), run<8, 3>(in, out, ns, 512),
           run<1, 4>(in, out, ns, 512), run<8, 4>(in, out, ns, 512), run<1, 5>(in, out, ns, 512), run<8, 5>(in, out, ns, 512));
    return 0;
}
