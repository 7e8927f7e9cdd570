// Microbenchmark: does the ORDER in which a persistent grid walks the packed
// input matter on MI355X?  Read-only, 48 bytes per lane per group (k_fir8's load
// shape), GPT groups per thread per tile, tile = 256*GPT groups.
//   order 0: block b owns the contiguous tile range [b*tpb, (b+1)*tpb)   (k_fir8)
//   order 1: tiles interleaved across blocks: b, b+G, b+2G, ...
//   order 2: contiguous ranges, but block b starts its range rotated by (b*7)%tpb tiles
//   order 3: blocks paired per XCD: tile index permuted so the 8 XCDs walk 8 separate fronts
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int GPT, int ORDER>
__global__ __launch_bounds__(256) void k(const uint4 *__restrict__ in, uint4 *__restrict__ out, int ntiles, int tpb)
{
    const int b = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int i = 0; i < tpb; ++i) {
        int t;
        if (ORDER == 0) t = b * tpb + i;
        else if (ORDER == 1) t = i * G + b;
        else if (ORDER == 2) t = b * tpb + (i + (b * 7) % tpb) % tpb;
        else { const int x = b & 7, r = b >> 3; t = x * (ntiles / 8) + i * (G / 8) + r; }
        if (t >= ntiles) continue;
        const uint4 *src = in + ((size_t)t * 256 * GPT + tid) * 3;
        uint4 v[GPT][3];
#pragma unroll
        for (int k = 0; k < GPT; ++k)
#pragma unroll
            for (int w = 0; w < 3; ++w) v[k][w] = src[(size_t)k * 256 * 3 + w];
#pragma unroll
        for (int k = 0; k < GPT; ++k)
#pragma unroll
            for (int w = 0; w < 3; ++w) { acc.x ^= v[k][w].x; acc.y += v[k][w].y; acc.z ^= v[k][w].z; acc.w += v[k][w].w; }
    }
    if (acc.x == 0x12345678u) out[b * 256 + tid] = acc;
}

template <int GPT, int ORDER>
static float run(const uint4 *in, uint4 *out, size_t ns, int blocks)
{
    const int ntiles = (int)(ns / 8 / (256 * GPT));
    const int tpb = (ntiles + blocks - 1) / blocks;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 8; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<GPT, ORDER>), dim3(blocks), dim3(256), 0, 0, in, out, ntiles, tpb);
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
    CHECK(hipMalloc(&out, 1 << 24));
    CHECK(hipMemset(in, 1, ns * 6));
    for (int blocks : { 512, 1024, 2048 }) {
        printf("blocks=%d GPT=2: contiguous %.3f  interleaved %.3f  rotated %.3f  xcd-fronts %.3f ms\n", blocks,
               run<2, 0>(in, out, ns, blocks), run<2, 1>(in, out, ns, blocks), run<2, 2>(in, out, ns, blocks), run<2, 3>(in, out, ns, blocks));
        printf("blocks=%d GPT=4: contiguous %.3f  interleaved %.3f  rotated %.3f  xcd-fronts %.3f ms\n", blocks,
               run<4, 0>(in, out, ns, blocks), run<4, 1>(in, out, ns, blocks), run<4, 2>(in, out, ns, blocks), run<4, 3>(in, out, ns, blocks));
    }
    return 0;
}
