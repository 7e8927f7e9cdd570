// Which pairs of streams feel the HBM "extent classes" (NOTEBOOK.md rounds 1-3 5 (o)-(r))?  One 160 GiB arena, 8 GiB slots; stream A
// in slot 0, stream B in every slot; three kernels over 1 GiB per stream: read A + write B (the DDC's case), read A +
// read B, write A + write B.  Prints ms per launch for every slot of B.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>   // 0: read A, write B   1: read A, read B   2: write A, write B
__global__ __launch_bounds__(256) void k(u32x4 *__restrict__ a, u32x4 *__restrict__ b, long long n16, u32x4 *sink)
{
    const long long stride = (long long)gridDim.x * 1024;
    u32x4 acc = { 0u, 0u, 0u, 0u };
    for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n16; i += stride) {
        u32x4 v[4], w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE != 2) v[u] = a[i + 256 * u];
            if (MODE == 1) w[u] = b[i + 256 * u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) __builtin_nontemporal_store(v[u], b + i + 256 * u);
            if (MODE == 1) acc += v[u] ^ w[u];
            if (MODE == 2) {
                const u32x4 c = { (unsigned)i, (unsigned)u, 3u, 4u };
                __builtin_nontemporal_store(c, a + i + 256 * u);
                __builtin_nontemporal_store(c, b + i + 256 * u);
            }
        }
    }
    if (MODE == 1 && acc.x == 0x12345678u) *sink = acc;
}

template <int MODE>
static float run(char *a, char *b, long long n16, u32x4 *sink)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<MODE>, dim3(4096), dim3(256), 0, 0, (u32x4 *)a, (u32x4 *)b, n16, sink);
    hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k<MODE>, dim3(4096), dim3(256), 0, 0, (u32x4 *)a, (u32x4 *)b, n16, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / 4;
}

int main(int argc, char **argv)
{
    const size_t GiB = (size_t)1 << 30, slot = 8 * GiB;
    const int nslot = argc > 1 ? atoi(argv[1]) : 20;
    char *arena; u32x4 *sink;
    CHECK(hipMalloc(&arena, nslot * slot));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(arena, 1, nslot * slot));
    const long long n16 = (long long)(GiB / 16);
    const char *name[3] = { "read A + write B", "read A + read B ", "write A + write B" };
    for (int mode = 0; mode < 3; ++mode) {
        printf("%s, A in slot 0, B in slot:", name[mode]);
        for (int o = 0; o < nslot; ++o) {
            char *a = arena, *b = arena + o * slot + 2 * GiB;
            float ms = mode == 0 ? run<0>(a, b, n16, sink) : mode == 1 ? run<1>(a, b, n16, sink) : run<2>(a, b, n16, sink);
            printf(" %.3f", ms);
        }
        printf("\n");
    }
    return 0;
}
